// One translation unit per (curve, S): instantiates the two NTT pass kernels and their launcher.
// Built with -DZK_CURVE_SEL=<0|1> -DZK_NTT_S=<3..9> (ark_plonk_amd/build.py).
#include "ntt_pass.cuh"

#if ZK_CURVE_SEL == 0
typedef Fu<FrBls12_381UParams> FrSel;
#define ZK_LAUNCHER_NAME2(s) zk_ntt_pass_c0_s##s
#else
typedef Fu<FrBn254UParams> FrSel;
#define ZK_LAUNCHER_NAME2(s) zk_ntt_pass_c1_s##s
#endif
#define ZK_LAUNCHER_NAME(s) ZK_LAUNCHER_NAME2(s)

extern "C" int ZK_LAUNCHER_NAME(ZK_NTT_S)(int final_pass, const NttPassArgs* a, hipStream_t st) {
    constexpr int S = ZK_NTT_S;
    // one wavefront per tile of 512 elements, four tiles per workgroup
    const unsigned waves = a->n_tiles < 4 ? a->n_tiles : 4;
    const unsigned blocks = (a->n_tiles + waves - 1) / waves;
    const unsigned polys = a->n_batch ? a->n_batch : 1;
    if (final_pass)
        hipLaunchKernelGGL((ntt_pass_final<FrSel, S>), dim3(blocks, polys), dim3(64 * waves), 0, st, *a);
    else
        hipLaunchKernelGGL((ntt_pass_mid<FrSel, S>), dim3(blocks, polys), dim3(64 * waves), 0, st, *a);
    return (int)hipGetLastError();
}
