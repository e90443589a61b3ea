// Timing sweep of rocPRIM onesweep configurations for the two sorts of the binning stage (gfx950).
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -DCFG_BS=.. -DCFG_IPT=.. -DCFG_BITS=.. -DCFG_ALGO=.. tools/tune/sort_tune.hip -o /tmp/sort_tune && /tmp/sort_tune
#include <hip/hip_runtime.h>
#include <rocprim/device/device_radix_sort.hpp>
#include <cstdio>
#include <vector>
#include <random>
#include <cstdint>

#ifndef CFG_BS
#define CFG_BS 0
#endif
#if CFG_BS
using OneSweep = rocprim::radix_sort_onesweep_config<rocprim::kernel_config<256, 12>, rocprim::kernel_config<CFG_BS, CFG_IPT>, CFG_BITS,
                                                     rocprim::block_radix_rank_algorithm::CFG_ALGO>;
using Config = rocprim::radix_sort_config<rocprim::default_config, rocprim::default_config, OneSweep>;
#else
using Config = rocprim::default_config;
#endif

template <class K>
static float run(size_t n, unsigned end_bit, unsigned maxkey)
{
    std::vector<K> hk(n);
    std::vector<uint32_t> hv(n);
    std::mt19937 rng(1);
    for (size_t i = 0; i < n; i++) { hk[i] = (K)(rng() % maxkey); hv[i] = (uint32_t)i; }
    K *ki, *ko; uint32_t *vi, *vo;
    hipMalloc(&ki, n * sizeof(K)); hipMalloc(&ko, n * sizeof(K)); hipMalloc(&vi, n * 4); hipMalloc(&vo, n * 4);
    hipMemcpy(ki, hk.data(), n * sizeof(K), hipMemcpyHostToDevice);
    hipMemcpy(vi, hv.data(), n * 4, hipMemcpyHostToDevice);
    size_t bytes = 0;
    rocprim::radix_sort_pairs<Config>(nullptr, bytes, ki, ko, vi, vo, n, 0u, end_bit);
    void* tmp; hipMalloc(&tmp, bytes);
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    for (int i = 0; i < 3; i++) rocprim::radix_sort_pairs<Config>(tmp, bytes, ki, ko, vi, vo, n, 0u, end_bit);
    hipDeviceSynchronize();
    hipEventRecord(a);
    const int R = 20;
    for (int i = 0; i < R; i++) rocprim::radix_sort_pairs<Config>(tmp, bytes, ki, ko, vi, vo, n, 0u, end_bit);
    hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    // stability / order check on the last run
    std::vector<K> ok(n); std::vector<uint32_t> ov(n);
    hipMemcpy(ok.data(), ko, n * sizeof(K), hipMemcpyDeviceToHost); hipMemcpy(ov.data(), vo, n * 4, hipMemcpyDeviceToHost);
    for (size_t i = 1; i < n; i++)
        if (ok[i - 1] > ok[i] || (ok[i - 1] == ok[i] && ov[i - 1] > ov[i])) { printf("  NOT SORTED/STABLE at %zu\n", i); break; }
    hipFree(ki); hipFree(ko); hipFree(vi); hipFree(vo); hipFree(tmp);
    return ms / R;
}

int main()
{
    printf("cfg bs=%d ipt=%d bits=%d : tile sort (u16,u32) 16.4M 13 bits: %.4f ms | depth sort (u32,u32) 3M 32 bits: %.4f ms\n", CFG_BS,
#if CFG_BS
           CFG_IPT, CFG_BITS,
#else
           0, 0,
#endif
           run<uint16_t>(16403154, 13, 8160), run<uint32_t>(3000000, 32, 0xffffffffu));
    return 0;
}
