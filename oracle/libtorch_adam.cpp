// TEST INFRASTRUCTURE (oracle/): the reference's optimiser itself -- libtorch's torch::optim::Adam, the class
// RawGaussianModel::initOptimizers instantiates (src/raw_gs_model.cpp:654-674) and optimizersStep() steps (:696-705) --
// wrapped for the parity tests, on CPU or GPU tensors.  Nothing of the algorithm is restated here: construction follows the
// reference call site (one optimizer per tensor; lr, eps and betas passed through `float` variables exactly as :661-664
// computes them: float eps = 1e-15 / sqrt(BS), float B1 = 1 - BS * (1 - 0.9), float B2 = 1 - BS * (1 - 0.999), BS = 1), the
// arithmetic is whatever the installed libtorch (2.10.0+rocm7.0) does.  Only tests/ may load this module.
#include <torch/extension.h>
#include <torch/optim/adam.h>

#include <cmath>
#include <memory>
#include <vector>

namespace {

struct RefAdam {
    std::vector<torch::Tensor> params;
    std::vector<std::unique_ptr<torch::optim::Adam>> opts;
    std::vector<double> lrs;

    RefAdam(std::vector<torch::Tensor> p, std::vector<double> lr) : lrs(std::move(lr)) {
        TORCH_CHECK(p.size() == lrs.size(), "one learning rate per tensor");
        for (auto& t : p) params.push_back(t.detach().clone().requires_grad_(true));
        init();
    }
    // RawGaussianModel::initOptimizers: delete + new for every tensor (state gone, step 0)
    void init() {
        opts.clear();
        float BS = 1;
        float eps = 1e-15 / std::sqrt(BS);
        float B1 = 1 - BS * (1 - 0.9);
        float B2 = 1 - BS * (1 - 0.999);
        for (size_t k = 0; k < params.size(); k++) {
            float lr = (float)lrs[k];   // (meansLr etc. are float members, raw_gs_model.h; AdamOptions takes a double)
            opts.emplace_back(new torch::optim::Adam({params[k]}, torch::optim::AdamOptions(lr).eps(eps).betas(std::make_tuple(B1, B2))));
        }
    }
    // loss.backward() leaves .grad; optimizersStep(); optimizersZeroGrad()
    void step(const std::vector<torch::Tensor>& grads) {
        TORCH_CHECK(grads.size() == params.size());
        for (size_t k = 0; k < params.size(); k++) params[k].mutable_grad() = grads[k].detach().clone();
        for (auto& o : opts) o->step();
        for (size_t k = 0; k < params.size(); k++) { opts[k]->zero_grad(); params[k].mutable_grad().reset(); }
    }
    std::vector<torch::Tensor> parameters() const {
        std::vector<torch::Tensor> out;
        for (auto& t : params) out.push_back(t.detach());
        return out;
    }
    std::vector<torch::Tensor> state(bool sq) const {
        std::vector<torch::Tensor> out;
        for (size_t k = 0; k < params.size(); k++) {
            auto& st = opts[k]->state();
            auto it = st.find(params[k].unsafeGetTensorImpl());
            if (it == st.end()) { out.push_back(torch::Tensor()); continue; }
            auto& s = static_cast<torch::optim::AdamParamState&>(*it->second);
            out.push_back(sq ? s.exp_avg_sq() : s.exp_avg());
        }
        return out;
    }
    // the scalars the reference hands to AdamOptions, as doubles: (eps, beta1, beta2)
    static std::vector<double> scalars() {
        float BS = 1;
        float eps = 1e-15 / std::sqrt(BS);
        float B1 = 1 - BS * (1 - 0.9);
        float B2 = 1 - BS * (1 - 0.999);
        return {(double)eps, (double)B1, (double)B2};
    }
};

}  // namespace

PYBIND11_MODULE(TORCH_EXTENSION_NAME, m) {
    pybind11::class_<RefAdam>(m, "RefAdam")
        .def(pybind11::init<std::vector<torch::Tensor>, std::vector<double>>())
        .def("init", &RefAdam::init)
        .def("step", &RefAdam::step)
        .def("parameters", &RefAdam::parameters)
        .def("exp_avg", [](const RefAdam& a) { return a.state(false); })
        .def("exp_avg_sq", [](const RefAdam& a) { return a.state(true); })
        .def_static("scalars", &RefAdam::scalars);
}
