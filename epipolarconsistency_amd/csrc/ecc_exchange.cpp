// ecc_exchange.hip -- host-side sum of one float64 per rank between the processes of one node (C ABI, no device
// code).
//
// The only exchange step of the path is the final sum over pairs (ref: EpipolarConsistencyRadonIntermediate.cpp:
// 216-224).  With the pair range sharded over one process per GPU every rank ends an evaluation with an 8-byte
// partial sum that is ALREADY in host memory (sum_pairs_kernel writes it to a pinned host address, the optimiser that
// consumes the value runs on the host).  Sending it back to the device for a collective launch costs more than the
// rank's whole pair kernel at 8 GPUs, so the scalar goes through a POSIX shared-memory segment instead: every rank
// stores {value, generation} into its own cache line, then reads the lines of all ranks in rank order and adds them
// up -- the same order on every rank, so all ranks return the same bits.  Bulk data (the Radon intermediates) still
// moves GPU-to-GPU with RCCL.
//
// Two generations of slots (g & 1): a rank can be at most one evaluation ahead of the slowest one, because it needs
// everyone's value of generation g to finish g.
#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

#include <atomic>
#include <chrono>
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <new>
#include <string>

#include "../../include/ecc_hip.h"

#define ECC_EXPORT extern "C" __attribute__((visibility("default")))

extern "C" int ecc_set_error(int code, const char* msg);  // ecc_capi.hip: records the message for ecc_last_error()

namespace {

struct alignas(128) Slot {
    std::atomic<uint64_t> generation;
    double value;
};

struct Segment {
    std::atomic<uint64_t> ready;  // written last by rank 0: world size
    uint64_t pad[15];
    Slot slots[2][ECC_EXCHANGE_MAX_RANKS];
};

inline void cpu_relax()
{
#if defined(__x86_64__) || defined(__i386__)
    __builtin_ia32_pause();
#endif
}

double timeout_seconds()
{
    const char* e = std::getenv("ECC_EXCHANGE_TIMEOUT_S");
    const double t = e ? std::atof(e) : 60.0;
    return t > 0 ? t : 60.0;
}

}  // namespace

struct ecc_exchange {
    Segment* seg = nullptr;
    std::string name;
    int rank = 0, world = 1;
    uint64_t generation = 0;
    double timeout_s = 60.0;
    bool failed = false;  // a sum timed out: the ranks no longer agree on the generation, every later sum fails
};

ECC_EXPORT int ecc_exchange_open(const char* name, int rank, int world, ecc_exchange** out)
{
    if (!name || !out || name[0] != '/') return ecc_set_error(ECC_ERR_INVALID_ARGUMENT, "exchange name must start with '/'");
    if (world < 1 || world > ECC_EXCHANGE_MAX_RANKS || rank < 0 || rank >= world)
        return ecc_set_error(ECC_ERR_INVALID_ARGUMENT, "rank/world outside [0, ECC_EXCHANGE_MAX_RANKS]");
    *out = nullptr;
    ecc_exchange* ex = new (std::nothrow) ecc_exchange;
    if (!ex) return ecc_set_error(ECC_ERR_OUT_OF_MEMORY, "out of host memory");
    ex->name = name;
    ex->rank = rank;
    ex->world = world;
    ex->timeout_s = timeout_seconds();
    int fd = -1;
    if (rank == 0) {
        shm_unlink(name);  // a stale segment of a crashed run
        fd = shm_open(name, O_CREAT | O_EXCL | O_RDWR, 0600);
        if (fd >= 0 && ftruncate(fd, sizeof(Segment)) != 0) {
            close(fd);
            shm_unlink(name);
            fd = -1;
        }
    } else {
        // rank 0 creates; the others poll until it exists and is initialised
        const auto t0 = std::chrono::steady_clock::now();
        for (;;) {
            fd = shm_open(name, O_RDWR, 0600);
            if (fd >= 0) {
                struct stat st;
                if (fstat(fd, &st) == 0 && (size_t)st.st_size >= sizeof(Segment)) break;
                close(fd);
                fd = -1;
            }
            if (std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() > ex->timeout_s) break;
            usleep(200);
        }
    }
    if (fd < 0) {
        delete ex;
        return ecc_set_error(ECC_ERR_UNSUPPORTED, "cannot open the shared-memory segment of the exchange");
    }
    void* p = mmap(nullptr, sizeof(Segment), PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
    close(fd);
    if (p == MAP_FAILED) {
        if (rank == 0) shm_unlink(name);
        delete ex;
        return ecc_set_error(ECC_ERR_UNSUPPORTED, "cannot map the shared-memory segment of the exchange");
    }
    ex->seg = static_cast<Segment*>(p);
    if (rank == 0) {
        std::memset(p, 0, sizeof(Segment));  // (fresh pages are zero already)
        ex->seg->ready.store((uint64_t)world, std::memory_order_release);
    } else {
        const auto t0 = std::chrono::steady_clock::now();
        while (ex->seg->ready.load(std::memory_order_acquire) != (uint64_t)world) {
            if (std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() > ex->timeout_s) {
                munmap(p, sizeof(Segment));
                delete ex;
                return ecc_set_error(ECC_ERR_UNSUPPORTED, "exchange segment was not initialised by rank 0 (or for another world size)");
            }
            cpu_relax();
        }
    }
    *out = ex;
    return ECC_OK;
}

ECC_EXPORT int ecc_exchange_sum(ecc_exchange* ex, double partial, double* total)
{
    if (!ex || !total) return ecc_set_error(ECC_ERR_INVALID_ARGUMENT, "null argument");
    // After a timeout this rank has published a generation the others may never complete; retrying would publish the
    // next one into the other slot set and overwrite what slower ranks still wait for.  The exchange stays failed:
    // the caller tears the job down (or falls back to a collective on all ranks), it does not retry.
    if (ex->failed) return ecc_set_error(ECC_ERR_UNSUPPORTED, "exchange failed earlier (a rank timed out); close it");
    const uint64_t g = ++ex->generation;
    Slot* slots = ex->seg->slots[g & 1];
    slots[ex->rank].value = partial;
    slots[ex->rank].generation.store(g, std::memory_order_release);
    double sum = 0.0;
    std::chrono::steady_clock::time_point t0;
    bool timing = false;
    for (int r = 0; r < ex->world; ++r) {
        uint64_t spins = 0;
        while (slots[r].generation.load(std::memory_order_acquire) != g) {
            cpu_relax();
            if ((++spins & 0xfffff) == 0) {  // look at the clock every ~million polls only
                const auto now = std::chrono::steady_clock::now();
                if (!timing) {
                    t0 = now;
                    timing = true;
                } else if (std::chrono::duration<double>(now - t0).count() > ex->timeout_s) {
                    ex->failed = true;
                    return ecc_set_error(ECC_ERR_UNSUPPORTED, "exchange timed out waiting for another rank");
                }
            }
        }
        sum += slots[r].value;  // rank order on every rank: identical bits everywhere
    }
    *total = sum;
    return ECC_OK;
}

ECC_EXPORT int ecc_exchange_close(ecc_exchange* ex)
{
    if (!ex) return ECC_OK;
    if (ex->seg) munmap(ex->seg, sizeof(Segment));
    if (ex->rank == 0) shm_unlink(ex->name.c_str());  // mappings of the other ranks stay valid until they unmap
    delete ex;
    return ECC_OK;
}
