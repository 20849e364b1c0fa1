// placeholder until the fused path lands: the ABI falls back to the generic kernels.
#include "internal.h"
namespace pnp {
Fused256* fused256_create(int, hipError_t* err) { *err = hipErrorNotSupported; return nullptr; }
void fused256_destroy(Fused256*) {}
hipError_t fused256_prepare(Fused256*, hipStream_t, const float2*, const uint8_t*, const int32_t*, int) { return hipErrorNotSupported; }
hipError_t fused256_run(Fused256*, hipStream_t, float*, float*, float*, int, int, bool, float, ProxParams) { return hipErrorNotSupported; }
hipError_t fused256_dc(Fused256*, hipStream_t, const float*, const float*, float*, int, float) { return hipErrorNotSupported; }
int fused256_kernels_per_iteration() { return 2; }
}
