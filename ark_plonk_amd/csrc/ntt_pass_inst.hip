// One translation unit per (curve, S): instantiates the two NTT pass kernels and their launcher.
// Built with -DZK_CURVE_SEL=<0|1> -DZK_NTT_S=<3..10> (ark_plonk_amd/build.py).
#include "ntt_pass.cuh"

#if ZK_CURVE_SEL == 0
typedef Fp<FrBls12_381Params> FrSel;
#define ZK_LAUNCHER_NAME2(s) zk_ntt_pass_c0_s##s
#else
typedef Fp<FrBn254Params> FrSel;
#define ZK_LAUNCHER_NAME2(s) zk_ntt_pass_c1_s##s
#endif
#define ZK_LAUNCHER_NAME(s) ZK_LAUNCHER_NAME2(s)

extern "C" int ZK_LAUNCHER_NAME(ZK_NTT_S)(int final_pass, const NttPassArgs* a, uint64_t n_tiles, uint32_t threads, size_t shmem,
                                          hipStream_t st) {
    constexpr int S = ZK_NTT_S;
    hipError_t e = hipSuccess;
    if (final_pass) {
        if (shmem > 48 * 1024)
            e = hipFuncSetAttribute((const void*)ntt_pass_final<FrSel, S>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shmem);
        if (e != hipSuccess) return (int)e;
        hipLaunchKernelGGL((ntt_pass_final<FrSel, S>), dim3((unsigned)n_tiles), dim3(threads), shmem, st, *a);
    } else {
        if (shmem > 48 * 1024)
            e = hipFuncSetAttribute((const void*)ntt_pass_mid<FrSel, S>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shmem);
        if (e != hipSuccess) return (int)e;
        hipLaunchKernelGGL((ntt_pass_mid<FrSel, S>), dim3((unsigned)n_tiles), dim3(threads), shmem, st, *a);
    }
    return (int)hipGetLastError();
}
