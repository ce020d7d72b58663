// The adapter's NRRD-typed signatures -- the ones the reference's callers use (ref: RadonIntermediate.h:31-41,47,53,80-83;
// Gui/ComputeRadonIntermediate.hxx:75-83, Gui/InputDataRadonIntermediate.cpp:71) -- compiled against the reference's
// own header-only NRRD library where it lies (-I<reference>/code/HeaderOnly, oracle/Makefile target `adapter_nrrd`).
//   test_adapter_nrrd <image.nrrd> <n_alpha> <n_t> <dtr_from_python.nrrd> <out_dir>
// Prints results as text for tests/test_cpp_adapter.py.  Without arguments: exit code 2, no device touched.
#include <cstdio>
#include <cstdlib>

#include "EpipolarConsistencyHip.hxx"

#ifndef ECC_ADAPTER_HAVE_NRRD
#error "compile with the reference's HeaderOnly directory on the include path"
#endif

using namespace EpipolarConsistency;

int main(int argc, char** argv)
{
    if (argc < 6) return 2;
    const int n_alpha = atoi(argv[2]), n_t = atoi(argv[3]);
    const std::string out_dir = argv[5];
    try {
        NRRD::Image<float> img(argv[1]);
        if (!img) return 3;
        // ref: new RadonIntermediate(img, n_alpha, n_t, filter, post_process)  (Gui/ComputeRadonIntermediate.hxx:75)
        RadonIntermediate a(img, n_alpha, n_t, RadonIntermediate::Derivative, RadonIntermediate::Identity);
        a.readback();
        NRRD::ImageView<float>& d = a.data();  // ref: dtr->data() (ComputeRadonIntermediate.hxx:80)
        printf("computed %d %d %d %d %.9g\n", d.size(0), d.size(1), a.getOriginalImageSize(0), a.getOriginalImageSize(1), d[1234 % d.length()]);
        a.writePropertiesToMeta(d.meta_info);
        const std::string saved = out_dir + "/adapter_saved.nrrd";
        if (!d.save(saved)) return 4;
        // ref: new RadonIntermediate(path)  (Gui/InputDataRadonIntermediate.cpp:71)
        RadonIntermediate b(saved);
        b.readback();
        printf("reloaded %d %d %d %d %d %.9g\n", b.getRadonBinNumber(0), b.getRadonBinNumber(1), b.getOriginalImageSize(0),
               b.getOriginalImageSize(1), (int)b.getFilter(), b.data()[1234 % b.data().length()]);
        // a file written by the Python side (epipolarconsistency_amd/nrrd.py), Filter = None
        const std::string py_path = argv[4];
        RadonIntermediate p(py_path);
        p.readback();
        printf("python %d %d %d %d %d %.9g\n", p.getRadonBinNumber(0), p.getRadonBinNumber(1), p.getOriginalImageSize(0),
               p.getOriginalImageSize(1), (int)p.getFilter(), p.data()[77]);
        // ref: RadonIntermediate(const NRRD::ImageView<float>&) + replaceRadonIntermediateData + readPropertiesFromMeta
        RadonIntermediate c(d);
        c.readback();
        float line[3] = {0.6f, -0.8f, -30.0f};
        const float tex = c.tex2D(0.25f, 0.75f);
        const float smp = c.sample(line);
        printf("view %d %d %d %.9g %.9g %.9g %.9g\n", c.getRadonBinNumber(0), c.getRadonBinNumber(1), (int)c.isDerivative(), tex, smp,
               line[0], line[1]);
        NRRD::Image<float> twice;
        twice.clone(d);
        for (int k = 0; k < twice.length(); ++k) twice[k] *= 2.f;
        c.replaceRadonIntermediateData(twice);
        c.readback();
        printf("replaced %.9g %.9g\n", c.data()[1234 % c.data().length()], c.tex2D(0.25f, 0.75f));
        std::map<std::string, std::string> meta = twice.meta_info;
        meta["Filter"] = "None";
        c.readPropertiesFromMeta(meta);
        printf("remeta %d %d\n", (int)c.getFilter(), (int)c.isDerivative());
    } catch (const std::exception& e) {
        fprintf(stderr, "exception: %s\n", e.what());
        return 1;
    }
    return 0;
}
