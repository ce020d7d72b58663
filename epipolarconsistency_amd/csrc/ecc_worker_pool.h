// ecc_worker_pool.h -- the job hand-off of the single-process multi-GPU group (ecc_group.cpp), host-only.
//
// n ranks: rank 0 runs on the calling thread, ranks 1 .. n-1 on persistent workers that spin ~100 us for the next job
// (an optimiser calls back within that) and then sleep on a condition variable.  One job at a time; run_all returns
// when every rank has finished.  No HIP in here, so the hand-off is built and run under ThreadSanitizer on the CPU
// (scripts/sanitize.sh, tests/c/tsan_worker_pool.cpp).
#ifndef ECC_WORKER_POOL_H
#define ECC_WORKER_POOL_H

#include <atomic>
#include <chrono>
#include <condition_variable>
#include <cstdint>
#include <exception>
#include <functional>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

class EccWorkerPool {
public:
    // thread_init(rank): first thing a worker does (ecc_group: hipSetDevice); last_error(): message behind a non-zero
    // status, read on the thread that produced it
    void start(int n_ranks, std::function<void(int)> thread_init, std::function<std::string()> last_error)
    {
        n_ = n_ranks;
        thread_init_ = std::move(thread_init);
        last_error_ = std::move(last_error);
        rc_.assign(n_, 0);
        err_.assign(n_, std::string());
        for (int r = 1; r < n_; ++r) workers_.emplace_back([this, r] { worker_main(r); });
    }

    void stop()
    {
        {
            std::lock_guard<std::mutex> lk(mu_);
            quit_.store(true, std::memory_order_release);
        }
        cv_.notify_all();
        for (std::thread& t : workers_) t.join();
        workers_.clear();
    }

    ~EccWorkerPool()
    {
        if (!workers_.empty()) stop();
    }

    int size() const { return n_; }

    // Runs job(rank) for every rank; returns the first rank with a non-zero status (its status and message in
    // status(rank) / message(rank)), or -1.  exception_status: what a rank that threw is recorded with.
    int run_all(std::function<int(int)> job, int exception_status)
    {
        job_ = std::move(job);
        exception_status_ = exception_status;
        for (int r = 0; r < n_; ++r) {
            rc_[r] = 0;
            err_[r].clear();
        }
        if (n_ > 1) {
            remaining_.store(n_ - 1, std::memory_order_relaxed);
            {
                // the generation changes under the mutex so that a worker about to sleep cannot miss it
                std::lock_guard<std::mutex> lk(mu_);
                generation_.fetch_add(1, std::memory_order_release);
            }
            if (sleepers_.load() > 0) cv_.notify_all();
        }
        run_rank(0);
        if (n_ > 1) {
            unsigned spins = 0;
            while (remaining_.load(std::memory_order_acquire) != 0) {
                relax();
                if ((++spins & 0xffff) == 0) std::this_thread::yield();
            }
        }
        for (int r = 0; r < n_; ++r)
            if (rc_[r] != 0) return r;
        return -1;
    }

    int status(int rank) const { return rc_[rank]; }
    const std::string& message(int rank) const { return err_[rank]; }

private:
    static void relax()
    {
#if defined(__x86_64__) || defined(__i386__)
        __builtin_ia32_pause();
#endif
    }
    static double now_s() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

    void run_rank(int rank)
    {
        int rc = 0;
        try {
            rc = job_(rank);
        } catch (const std::exception& e) {
            err_[rank] = e.what();
            rc_[rank] = exception_status_;
            return;
        }
        rc_[rank] = rc;
        if (rc != 0 && last_error_) err_[rank] = last_error_();  // thread-local in the library: copy it out of the worker
    }

    void worker_main(int rank)
    {
        if (thread_init_) thread_init_(rank);
        uint64_t seen = 0;
        for (;;) {
            const double t0 = now_s();
            unsigned spins = 0;
            while (generation_.load(std::memory_order_acquire) == seen && !quit_.load(std::memory_order_acquire)) {
                relax();
                if ((++spins & 0xff) == 0 && now_s() - t0 > 100e-6) {
                    std::unique_lock<std::mutex> lk(mu_);
                    sleepers_.fetch_add(1);
                    cv_.wait(lk, [&] { return generation_.load() != seen || quit_.load(); });
                    sleepers_.fetch_sub(1);
                }
            }
            if (quit_.load(std::memory_order_acquire)) return;
            seen = generation_.load(std::memory_order_acquire);
            run_rank(rank);
            remaining_.fetch_sub(1, std::memory_order_release);
        }
    }

    int n_ = 0;
    std::vector<std::thread> workers_;
    std::mutex mu_;
    std::condition_variable cv_;
    std::atomic<uint64_t> generation_{0};
    std::atomic<int> remaining_{0};
    std::atomic<int> sleepers_{0};
    std::atomic<bool> quit_{false};
    std::function<int(int)> job_;
    std::function<void(int)> thread_init_;
    std::function<std::string()> last_error_;
    int exception_status_ = -1;
    std::vector<int> rc_;
    std::vector<std::string> err_;
};

#endif
