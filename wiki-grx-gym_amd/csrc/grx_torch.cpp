// grx_torch.cpp -- zero-copy torch views over library-owned buffers.
//
// Native successor of the reference's only native source, gymtorch.cpp:33-158 (gtWrapTensor:
// raw device/host pointer -> non-owning torch::Tensor, contiguous or strided).  Differences:
// strides are explicit (SoA state is exposed as (N, k) with strides (1, N)), errors throw instead
// of printf + empty tensor (gymtorch.cpp:40-51), and the device is a HIP device ("cuda" in
// torch-ROCm).  The tensor never owns or frees the memory (gymtorch.py:105 own_data=False).
#include <torch/extension.h>

#include <vector>

static torch::Tensor wrap(int64_t ptr, int64_t dtype, std::vector<int64_t> shape, std::vector<int64_t> strides,
                          int64_t device_index) {
    TORCH_CHECK(ptr != 0, "grx wrap: null data pointer");
    TORCH_CHECK(shape.size() == strides.size() && !shape.empty(), "grx wrap: shape/stride rank mismatch");
    c10::ScalarType st;
    switch (dtype) {  // grx_dtype (include/grx.h)
    case 0: st = torch::kFloat32; break;
    case 1: st = torch::kUInt8; break;
    case 2: st = torch::kInt32; break;
    case 3: st = torch::kInt64; break;
    default: TORCH_CHECK(false, "grx wrap: unknown dtype ", dtype);
    }
    auto opts = torch::TensorOptions().dtype(st).requires_grad(false);
    opts = device_index < 0 ? opts.device(torch::kCPU) : opts.device(torch::Device(torch::kCUDA, (c10::DeviceIndex)device_index));
    return torch::from_blob(reinterpret_cast<void*>(ptr), shape, strides, [](void*) {}, opts);
}

PYBIND11_MODULE(TORCH_EXTENSION_NAME, m) {
    m.def("wrap", &wrap, "non-owning tensor view of a grx buffer (ptr, grx_dtype, shape, strides, device index or -1)");
}
