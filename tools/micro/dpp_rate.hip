// Issue rate of DPP-modified VALU instructions against plain ones on gfx950 (round 5: why the label-list sort networks cost what they cost).
// hipcc --offload-arch=gfx950 -O3 tools/micro/dpp_rate.hip -o tools/micro/dpp_rate && tools/micro/dpp_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
template <int MODE>
__global__ void __launch_bounds__(256) k(uint32_t* out, int iters) {
    uint32_t a = threadIdx.x * 2654435761u + blockIdx.x, b = a ^ 0x9e3779b9u, c = a + 77u, d = b + 99u;
    for (int i = 0; i < iters; i++) {
#pragma unroll
        for (int u = 0; u < 16; u++) {
            if (MODE == 0) {            // plain: four independent chains of v_min_u32 / v_max_u32
                a = min(a, b + u); b = max(b, c); c = min(c, d + u); d = max(d, a);
            } else if (MODE == 1) {     // the same with a quad_perm DPP source
                a = min(a, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)b, 0xB1, 0xf, 0xf, true) + u);
                b = max(b, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)c, 0x4E, 0xf, 0xf, true));
                c = min(c, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)d, 0xB1, 0xf, 0xf, true) + u);
                d = max(d, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)a, 0x4E, 0xf, 0xf, true));
            } else if (MODE == 2) {     // row_ror:8
                a = min(a, (uint32_t)__builtin_amdgcn_update_dpp((int)b, (int)b, 0x128, 0xf, 0xf, false) + u);
                b = max(b, (uint32_t)__builtin_amdgcn_update_dpp((int)c, (int)c, 0x128, 0xf, 0xf, false));
                c = min(c, (uint32_t)__builtin_amdgcn_update_dpp((int)d, (int)d, 0x128, 0xf, 0xf, false) + u);
                d = max(d, (uint32_t)__builtin_amdgcn_update_dpp((int)a, (int)a, 0x128, 0xf, 0xf, false));
            } else {                    // ds_swizzle xor 16
                a = min(a, (uint32_t)__builtin_amdgcn_ds_swizzle((int)b, 0x401F) + u);
                b = max(b, (uint32_t)__builtin_amdgcn_ds_swizzle((int)c, 0x401F));
                c = min(c, (uint32_t)__builtin_amdgcn_ds_swizzle((int)d, 0x401F) + u);
                d = max(d, (uint32_t)__builtin_amdgcn_ds_swizzle((int)a, 0x401F));
            }
        }
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = a ^ b ^ c ^ d;
}
int main() {
    uint32_t* o; hipMalloc(&o, 4096 * 256 * 4);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int iters = 2000;
    const char* names[4] = {"plain min/max", "quad_perm DPP", "row_ror:8 DPP", "ds_swizzle"};
    for (int m = 0; m < 4; m++) {
        for (int rep = 0; rep < 2; rep++) {
            hipEventRecord(e0);
            if (m == 0) k<0><<<4096, 256>>>(o, iters); else if (m == 1) k<1><<<4096, 256>>>(o, iters); else if (m == 2) k<2><<<4096, 256>>>(o, iters); else k<3><<<4096, 256>>>(o, iters);
            hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            if (rep) printf("%-16s %8.3f ms  %.1f cross-lane ops per ns device-wide (4096 x 4 waves x %d x 64 ops)\n", names[m], ms, 4096.0 * 4 * iters * 64 / (ms * 1e6), iters);
        }
    }
    return 0;
}
