// Micro-benchmark: how many cycles does a SIMD of gfx950 spend per wave64 vector instruction, as a function of the waves per SIMD
// and of the instruction kind (v_fma_f32, v_add_f32_dpp, v_exp_f32, mix with LDS broadcast reads as in the blend kernels)?
//   hipcc --offload-arch=gfx950 -O3 -o valu_rate tools/tune/valu_rate.hip && ./valu_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

template <int KIND>
__global__ void __launch_bounds__(1024) rate_kernel(float* out, int iters, unsigned long long* cyc)
{
    __shared__ float4 s_rec[64];
    if (threadIdx.x < 64) s_rec[threadIdx.x] = make_float4(threadIdx.x * 1e-3f, 1.f, 0.5f, 0.25f);
    __syncthreads();
    float a[8];
    for (int k = 0; k < 8; k++) a[k] = threadIdx.x * 1e-6f + k;
    const float b = 1.0000001f, c = 1e-7f;
    const unsigned long long t0 = __builtin_readcyclecounter();
    for (int i = 0; i < iters; i++) {
        if (KIND == 0) {            // 64 independent-ish v_fma_f32 (8 chains)
#pragma unroll
            for (int r = 0; r < 8; r++)
#pragma unroll
                for (int k = 0; k < 8; k++) a[k] = fmaf(a[k], b, c);
        } else if (KIND == 1) {     // 64 v_add_f32_dpp (row_half_mirror)
#pragma unroll
            for (int r = 0; r < 8; r++)
#pragma unroll
                for (int k = 0; k < 8; k++)
                    asm volatile("v_add_f32_dpp %0, %0, %0 row_half_mirror row_mask:0xf bank_mask:0xf" : "+v"(a[k]));
        } else if (KIND == 2) {     // 56 fma + 8 exp
#pragma unroll
            for (int r = 0; r < 7; r++)
#pragma unroll
                for (int k = 0; k < 8; k++) a[k] = fmaf(a[k], b, c);
#pragma unroll
            for (int k = 0; k < 8; k++) a[k] = __builtin_amdgcn_exp2f(a[k] * 1e-3f);
        } else if (KIND == 3) {     // the blend kernels' shape: per 50 fma, two broadcast ds_read_b128 + one b32
#pragma unroll
            for (int g = 0; g < 2; g++) {
                const float4 r0 = s_rec[(i + g) & 63], r1 = s_rec[(i + g + 7) & 63];
                const float r2 = s_rec[(i + g + 13) & 63].x;
#pragma unroll
                for (int r = 0; r < 6; r++)
#pragma unroll
                    for (int k = 0; k < 8; k++) a[k] = fmaf(a[k], r == 0 ? r0.x : b, r == 1 ? r1.y : c);
                a[0] += r2 + r0.y + r0.z + r0.w + r1.x + r1.z + r1.w;
            }
        }
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    float s = 0;
    for (int k = 0; k < 8; k++) s += a[k];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

template <int KIND> void run(const char* name, int n_instr_per_iter)
{
    float* out; unsigned long long* cyc;
    hipMalloc(&out, 256 * 1024 * sizeof(float) * 4);
    hipMalloc(&cyc, 4096 * sizeof(unsigned long long));
    const int iters = 2000;
    for (int waves_per_simd : { 1, 2, 3, 4, 5, 8 }) {
        const int threads = 64 * 4 * waves_per_simd;          // one workgroup per CU: waves spread over the 4 SIMDs
        if (threads > 1024) {                                   // two workgroups per CU
            hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
            rate_kernel<KIND><<<512, threads / 2, 0, 0>>>(out, iters, cyc);
            hipDeviceSynchronize();
            hipEventRecord(e0); rate_kernel<KIND><<<512, threads / 2, 0, 0>>>(out, iters, cyc); hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            const double inst = (double)iters * n_instr_per_iter * waves_per_simd;       // per SIMD
            printf("%-34s %d waves/SIMD: %.3f ms -> %.2f ns per instruction per SIMD (x clock GHz = cycles)\n", name, waves_per_simd, ms, ms * 1e6 / inst);
            continue;
        }
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        rate_kernel<KIND><<<256, threads, 0, 0>>>(out, iters, cyc);
        hipDeviceSynchronize();
        hipEventRecord(e0); rate_kernel<KIND><<<256, threads, 0, 0>>>(out, iters, cyc); hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        std::vector<unsigned long long> h(256);
        hipMemcpy(h.data(), cyc, 256 * sizeof(unsigned long long), hipMemcpyDeviceToHost);
        const double inst = (double)iters * n_instr_per_iter * waves_per_simd;
        printf("%-34s %d waves/SIMD: %.3f ms -> %.2f ns per instruction per SIMD; s_memtime ticks per instr (wave 0) %.2f\n", name, waves_per_simd, ms,
               ms * 1e6 / inst, (double)h[0] / ((double)iters * n_instr_per_iter));
    }
    hipFree(out); hipFree(cyc);
}

int main()
{
    run<0>("v_fma_f32 x64", 64);
    run<1>("v_add_f32_dpp x64", 64);
    run<2>("56 fma + 8 (mul + v_exp_f32)", 72);
    run<3>("2 x (48 fma + 2 ds_read_b128 + b32)", 2 * (48 + 7));
    return 0;
}
