// Exercises the C++ adapter the way the reference's callers use the classes
// (ref: Gui/SingleImageMotion.h:84-90 -- overwrite one projection matrix, setProjectionMatrices, evaluate).
// Reads a raw float32 image stack + matrices written by the pytest driver, prints results as text.
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <set>
#include <vector>

#include "EpipolarConsistencyHip.hxx"

using namespace EpipolarConsistency;

int main(int argc, char** argv)
{
    if (argc == 4 && !strcmp(argv[1], "ompl")) {  // projection-table round trip (no device): load, report, save
        std::map<std::string, std::string> meta;
        std::vector<ProjectionMatrix> Ps = ProjTable::loadProjectionsOneMatrixPerLine(argv[2], &meta);
        printf("matrices %d\n", (int)Ps.size());
        for (std::map<std::string, std::string>::const_iterator it = meta.begin(); it != meta.end(); ++it)
            printf("meta %s=%s\n", it->first.c_str(), it->second.c_str());
        const int det[2] = {640, 480};
        return ProjTable::saveProjectionsOneMatrixPerLine(Ps, argv[3], meta["comment"], 0.308, det) ? 0 : 3;
    }
    if (argc < 8) return 2;
    const char* path = argv[1];
    const int n = atoi(argv[2]), n_u = atoi(argv[3]), n_v = atoi(argv[4]), n_alpha = atoi(argv[5]), n_t = atoi(argv[6]);
    const char* ppath = argv[7];
    std::vector<float> imgs((size_t)n * n_u * n_v);
    std::vector<double> Pflat(12 * (size_t)n);
    FILE* f = fopen(path, "rb");
    if (!f || fread(imgs.data(), sizeof(float), imgs.size(), f) != imgs.size()) return 3;
    fclose(f);
    f = fopen(ppath, "rb");
    if (!f || fread(Pflat.data(), sizeof(double), Pflat.size(), f) != Pflat.size()) return 3;
    fclose(f);
    try {
        std::vector<RadonIntermediate*> dtrs;
        std::vector<ProjectionMatrix> Ps(n);
        for (int k = 0; k < n; ++k) {
            dtrs.push_back(new RadonIntermediate(imgs.data() + (size_t)k * n_u * n_v, n_u, n_v, n_alpha, n_t,
                                                 RadonIntermediate::Derivative, RadonIntermediate::Identity));
            for (int e = 0; e < 12; ++e) Ps[k].data()[e] = Pflat[12 * k + e];
        }
        MetricRadonIntermediate ecc(Ps, dtrs);
        std::vector<float> cost((size_t)n * n, -1.f);
        printf("radius %.17g\n", ecc.getObjectRadius());
        {   // the free functions of EpipolarConsistency.h:36-46
            const std::vector<double> O = estimateIsoCenter(Ps);
            const std::pair<double, double> range = estimateAngularRange(Ps[0], Ps[2], 50.0);
            printf("freefn %.17g %.17g %.17g %.17g %.17g %.17g %.17g %.17g\n", estimateObjectRadius(Ps[0], n_u, n_v),
                   estimateAngularStep(Ps[0], Ps[2], n_u, n_v), range.first, range.second, O[0], O[1], O[2], O[3]);
        }
        printf("mean %.17g\n", ecc.evaluate(cost.data()));
        printf("cost10 %.9g cost01 %.9g\n", cost[0 + 1 * n], cost[1 + 0 * n]);
        std::set<int> views;
        views.insert(0); views.insert(2); views.insert(3);
        printf("subset %.17g\n", ecc.evaluate(views));
        dtrs[1]->readback();
        printf("dtr1 %zu %.9g bins %d %d size %d %d step %.17g\n", dtrs[1]->data().size(), dtrs[1]->data()[1234 % dtrs[1]->data().size()],
               dtrs[1]->getRadonBinNumber(0), dtrs[1]->getRadonBinNumber(1), dtrs[1]->getOriginalImageSize(0),
               dtrs[1]->getOriginalImageSize(1), dtrs[1]->getRadonBinSize(1));
        {   // host sampling helpers of the dtr class (RadonIntermediate.h:86-108)
            float line[3] = {0.6f, -0.8f, -30.0f};
            const float tex = dtrs[1]->tex2D(0.25f, 0.75f);
            const float smp = dtrs[1]->sample(line);
            printf("hostsample %.9g %.9g %.9g %.9g\n", tex, smp, line[0], line[1]);
        }
        std::vector<float> rs0, rs1, kappas;
        std::vector<std::pair<float, float> > loc0, loc1;
        const double pair_ecc = ecc.evaluateForImagePair(0, 2, &rs0, &rs1, &kappas, &loc0, &loc1);
        printf("pair02 %.17g %zu %zu %zu %.9g %.9g\n", pair_ecc, rs0.size(), kappas.size(), loc1.size(), kappas[0], loc0[0].first);
        ecc.setObjectRadius(50.0);
        printf("mean_r50 %.17g\n", ecc.evaluate());
        {   // the optimiser pattern (Gui/SingleImageMotion.h:84-90): one matrix overwritten per call, with and without the
            // pose-delta mode -- the same bits
            ecc.setObjectRadius(0.0);
            std::vector<ProjectionMatrix> moved = Ps;
            double full_values[3], inc_values[3];
            for (int pass = 0; pass < 2; ++pass) {
                ecc.setIncremental(pass == 1);
                ecc.setProjectionMatrices(Ps);
                ecc.evaluate();
                for (int step = 0; step < 3; ++step) {
                    moved = Ps;
                    moved[2].data()[9] += 0.25 * (step + 1);  // the translation column of view 2
                    ecc.setProjectionMatrices(moved);
                    (pass ? inc_values : full_values)[step] = ecc.evaluate();
                }
            }
            ecc.setIncremental(false);
            ecc.setProjectionMatrices(Ps);
            printf("incremental %d %.17g %.17g\n", (full_values[0] == inc_values[0] && full_values[1] == inc_values[1] &&
                   full_values[2] == inc_values[2] && full_values[0] != full_values[1]) ? 1 : 0, full_values[2], inc_values[2]);
        }
        {   // round 4: the one-launch path on / off and a batch of poses -- the same bits as one call per pose
            std::vector<std::vector<ProjectionMatrix> > poses;
            for (int step = 0; step < 4; ++step) {
                std::vector<ProjectionMatrix> moved = Ps;
                moved[1].data()[10] += 0.2 * (step + 1);
                poses.push_back(moved);
            }
            double one_by_one[4], small_off[4];
            for (int pass = 0; pass < 2; ++pass) {
                ecc.setSmallEval(pass == 0);
                for (int step = 0; step < 4; ++step) {
                    ecc.setProjectionMatrices(poses[step]);
                    (pass ? small_off : one_by_one)[step] = ecc.evaluate();
                }
            }
            ecc.setSmallEval(true);
            const std::vector<double> batch = ecc.evaluatePoses(poses);
            bool same = batch.size() == 4;
            for (int step = 0; same && step < 4; ++step) same = batch[step] == one_by_one[step] && small_off[step] == one_by_one[step];
            ecc.setProjectionMatrices(Ps);
            printf("round4 %d %.17g\n", (same && one_by_one[0] != one_by_one[1]) ? 1 : 0, batch[3]);
            // round 6: the same four poses as deltas of the current matrices (one moved view each) -- the same bits again, and
            // the current matrices stay
            std::vector<std::vector<int> > moved_views(4, std::vector<int>(1, 1));
            std::vector<std::vector<ProjectionMatrix> > moved_Ps;
            for (int step = 0; step < 4; ++step) moved_Ps.push_back(std::vector<ProjectionMatrix>(1, poses[step][1]));
            const double before = ecc.evaluate();
            const std::vector<double> deltas = ecc.evaluatePoseDeltas(moved_views, moved_Ps);
            bool same6 = deltas.size() == 4 && ecc.evaluate() == before;
            for (int step = 0; same6 && step < 4; ++step) same6 = deltas[step] == one_by_one[step];
            printf("round6 %d %.17g\n", same6 ? 1 : 0, deltas[3]);
        }
        {   // PreProccess (Gui/PreProccess.h): the two image calls one after the other on image 1, the fused stack call on
            // all images -- results written next to the input file for the driver to compare
            PreProccess pre;
            pre.intensity.scale = 0.5;
            pre.intensity.bias = 0.125;
            pre.border.zero[0] = 3;
            pre.image_geometry.flip_u = true;
            std::vector<float> one(imgs.begin() + (size_t)n_u * n_v, imgs.begin() + 2 * (size_t)n_u * n_v), all(imgs);
            pre.process(one.data(), n_u, n_v);
            pre.apply_weight_cos_principal_ray(one.data(), n_u, n_v, Ps[1]);
            pre.process_and_weight(all.data(), n, n_u, n_v, Ps);
            const bool same = std::memcmp(one.data(), all.data() + (size_t)n_u * n_v, sizeof(float) * n_u * n_v) == 0;
            FILE* o = fopen((std::string(path) + ".pre").c_str(), "wb");
            if (!o || fwrite(all.data(), sizeof(float), all.size(), o) != all.size()) return 3;
            fclose(o);
            printf("preprocess %d\n", same ? 1 : 0);
        }
        {
            MetricDirect direct(Ps, imgs.data(), n, n_u, n_v);
            std::vector<float> d0, d1, dk;
            const double pair = direct.evaluateForImagePair(0, 2, &d0, &d1, &dk);
            printf("direct %.17g %.17g %zu\n", direct.evaluate(), pair, dk.size());
        }
        for (size_t k = 0; k < dtrs.size(); ++k) delete dtrs[k];
    } catch (const std::exception& e) {
        fprintf(stderr, "exception: %s\n", e.what());
        return 1;
    }
    return 0;
}
