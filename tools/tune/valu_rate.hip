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
    const unsigned long long msk = __builtin_amdgcn_readfirstlane(iters) > 5 ? 0x5555aaaa5555aaaaull : 0ull;
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
        } else if (KIND == 4) {     // 64 plain v_fma_f32, forced (KIND 0 is SLP-packed into 32 v_pk_fma_f32 by -O3: check the ISA)
#pragma unroll
            for (int r = 0; r < 8; r++)
#pragma unroll
                for (int k = 0; k < 8; k++) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[k]) : "v"(b), "v"(c));
        } else if (KIND == 5) {     // 32 v_pk_fma_f32 on register pairs = 64 fp32 FMAs
#pragma unroll
            for (int r = 0; r < 8; r++)
#pragma unroll
                for (int k = 0; k < 4; k++) {
                    float2 v = make_float2(a[2 * k], a[2 * k + 1]);
                    const float2 bb = make_float2(b, b), cc = make_float2(c, c);
                    asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(v) : "v"(bb), "v"(cc));
                    a[2 * k] = v.x; a[2 * k + 1] = v.y;
                }
        } else if (KIND == 6) {     // 64 plain v_mul_f32 / v_add_f32 alternating, forced
#pragma unroll
            for (int r = 0; r < 4; r++)
#pragma unroll
                for (int k = 0; k < 8; k++) {
                    asm volatile("v_mul_f32 %0, %0, %1" : "+v"(a[k]) : "v"(b));
                    asm volatile("v_add_f32 %0, %0, %1" : "+v"(a[k]) : "v"(c));
                }
        } else if (KIND == 7) {     // 32 v_pk_mul_f32 / v_pk_add_f32 alternating
#pragma unroll
            for (int r = 0; r < 4; r++)
#pragma unroll
                for (int k = 0; k < 4; k++) {
                    float2 v = make_float2(a[2 * k], a[2 * k + 1]);
                    const float2 bb = make_float2(b, b), cc = make_float2(c, c);
                    asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(v) : "v"(bb));
                    asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(v) : "v"(cc));
                    a[2 * k] = v.x; a[2 * k + 1] = v.y;
                }
        } else if (KIND == 8) {     // 64 v_exp_f32, forced
#pragma unroll
            for (int r = 0; r < 8; r++)
#pragma unroll
                for (int k = 0; k < 8; k++) asm volatile("v_exp_f32 %0, %0" : "+v"(a[k]));
        } else if (KIND == 9) {     // 64 v_cndmask / v_cmp pairs (compare + select, as the blend kernels' skip masks)
#pragma unroll
            for (int r = 0; r < 4; r++)
#pragma unroll
                for (int k = 0; k < 8; k++) {
                    asm volatile("v_cmp_lt_f32 vcc, %0, %1" : : "v"(a[k]), "v"(b) : "vcc");
                    asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(a[k]) : "v"(c) : "vcc");
                }
        } else if (KIND == 10) {    // 64 v_cndmask_b32 alone (mask in an SGPR pair set before the loop)
#pragma unroll
            for (int r = 0; r < 8; r++)
#pragma unroll
                for (int k = 0; k < 8; k++) asm volatile("v_cndmask_b32 %0, %0, %1, %2" : "+v"(a[k]) : "v"(c), "s"(msk));
        } else if (KIND == 11) {    // 64 v_cmp_lt_f32 alone (writes vcc)
#pragma unroll
            for (int r = 0; r < 8; r++)
#pragma unroll
                for (int k = 0; k < 8; k++) asm volatile("v_cmp_lt_f32 vcc, %0, %1" : : "v"(a[k]), "v"(b) : "vcc");
        } else if (KIND == 12) {    // 32 v_permlane16_swap (two registers each)
#pragma unroll
            for (int r = 0; r < 8; r++)
#pragma unroll
                for (int k = 0; k < 4; k++) asm volatile("v_permlane16_swap_b32 %0, %1" : "+v"(a[2 * k]), "+v"(a[2 * k + 1]));
        } else if (KIND == 13) {    // 32 v_permlane32_swap
#pragma unroll
            for (int r = 0; r < 8; r++)
#pragma unroll
                for (int k = 0; k < 4; k++) asm volatile("v_permlane32_swap_b32 %0, %1" : "+v"(a[2 * k]), "+v"(a[2 * k + 1]));
        } else if (KIND == 14) {    // 64 v_mov_b32_dpp quad_perm
#pragma unroll
            for (int r = 0; r < 8; r++)
#pragma unroll
                for (int k = 0; k < 8; k++)
                    asm volatile("v_mov_b32_dpp %0, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf" : "+v"(a[k]));
        } else if (KIND == 15) {    // 32 x (ds_swizzle_b32 + v_add_f32): the cross-lane move on the LDS pipe
#pragma unroll
            for (int r = 0; r < 4; r++) {
                float t[8];
#pragma unroll
                for (int k = 0; k < 8; k++) asm volatile("ds_swizzle_b32 %0, %1 offset:swizzle(SWAP,1)" : "=v"(t[k]) : "v"(a[k]));
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
                for (int k = 0; k < 8; k++) asm volatile("v_add_f32 %0, %0, %1" : "+v"(a[k]) : "v"(t[k]));
            }
        } else if (KIND == 16) {    // 48 plain fma + 16 ds_swizzle in their shadow (does the LDS pipe take VALU issue time?)
            float t[8];
#pragma unroll
            for (int k = 0; k < 8; k++) asm volatile("ds_swizzle_b32 %0, %1 offset:swizzle(SWAP,1)" : "=v"(t[k]) : "v"(a[k]));
#pragma unroll
            for (int r = 0; r < 6; r++)
#pragma unroll
                for (int k = 0; k < 8; k++) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[k]) : "v"(b), "v"(c));
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
            for (int k = 0; k < 8; k++) asm volatile("v_add_f32 %0, %0, %1" : "+v"(a[k]) : "v"(t[k]));
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
    run<0>("fmaf x64 (-O3 packs: 32 v_pk_fma)", 64);
    run<4>("plain v_fma_f32 x64 (asm)", 64);
    run<5>("v_pk_fma_f32 x32 (per pk instr)", 32);
    run<6>("plain v_mul/v_add x64 (asm)", 64);
    run<7>("v_pk_mul/v_pk_add x32 (per pk)", 32);
    run<8>("v_exp_f32 x64 (asm)", 64);
    run<9>("v_cmp + v_cndmask x64 (asm)", 64);
    run<10>("v_cndmask_b32 (sgpr mask) x64", 64);
    run<11>("v_cmp_lt_f32 -> vcc x64", 64);
    run<12>("v_permlane16_swap x32 (per swap)", 32);
    run<13>("v_permlane32_swap x32 (per swap)", 32);
    run<14>("v_mov_b32_dpp x64", 64);
    run<15>("32 x (ds_swizzle + v_add) per pair", 32);
    run<16>("48 fma + 8 ds_swizzle + 8 add (per VALU)", 56);
    run<1>("v_add_f32_dpp x64", 64);
    run<2>("56 fma + 8 (mul + v_exp_f32)", 72);
    run<3>("2 x (48 fma + 2 ds_read_b128 + b32)", 2 * (48 + 7));
    return 0;
}
