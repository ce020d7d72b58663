// Syntax / type check of the adapter's Eigen branch (g++ -fsyntax-only -Wall -Werror -DECC_TEST_MOCK_EIGEN with
// tests/cpp/mock_eigen on the include path; never linked, never run): a caller written the way the reference's own callers
// are -- the objective of Gui/SingleImageMotion.h:84-90 (replace one view's matrix, setProjectionMatrices, evaluate) and the
// index-list overload evaluate(const std::vector<Eigen::Vector4i>&, float*) of EpipolarConsistencyRadonIntermediate.h:58.
#include "EpipolarConsistencyHip.hxx"

#ifndef ECC_ADAPTER_HAVE_EIGEN
#error "the adapter did not take its Eigen branch"
#endif

namespace {

struct OneViewObjective {
    std::vector<Geometry::ProjectionMatrix> Ps;
    std::vector<EpipolarConsistency::RadonIntermediate*> dtrs;
    std::vector<Eigen::Vector4i> indices;
    std::vector<float> tmp_results;
    EpipolarConsistency::MetricRadonIntermediate* ecc;
    int input_index;

    OneViewObjective() : ecc(0x0), input_index(0) {}

    double evaluate(const Geometry::ProjectionMatrix& P_input)
    {
        Ps[input_index] = P_input;
        if (!ecc) ecc = new EpipolarConsistency::MetricRadonIntermediate(Ps, dtrs);
        else ecc->setProjectionMatrices(Ps);
        return ecc->evaluate();
    }

    double evaluate_pairs_of_input_view()
    {
        indices.clear();
        for (int j = 0; j < (int)dtrs.size(); ++j) {
            if (j == input_index) continue;
            Eigen::Vector4i t;
            t[0] = input_index; t[1] = j; t[2] = input_index; t[3] = j;
            indices.push_back(t);
        }
        tmp_results.resize(indices.size());
        return ecc->evaluate(indices, tmp_results.data());
    }
};

double use(OneViewObjective& o)
{
    Geometry::ProjectionMatrix P;
    P(0, 0) = 1; P(1, 1) = 1; P(2, 3) = 1;
    const double* raw = P.data();
    EpipolarConsistency::PreProccess pre;
    pre.border.zero = EpipolarConsistency::PreProccess::constant4(0);
    pre.border.blanks.push_back(Eigen::Vector4i::Constant(2));
    float image[16] = {0};
    pre.process(image, 4, 4);
    pre.apply_weight_cos_principal_ray(image, 4, 4, P);
    return o.evaluate(P) + o.evaluate_pairs_of_input_view() + raw[0] + pre.border.zero[1];
}

}  // namespace

int main() { return sizeof(&use) ? 0 : 1; }
