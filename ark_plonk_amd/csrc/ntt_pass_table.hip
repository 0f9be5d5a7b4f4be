// Launcher lookup for the per-(curve, S) NTT pass objects (ntt_pass_inst.hip).
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stddef.h>

struct NttPassArgs;
typedef int (*NttPassLauncher)(int, const NttPassArgs*, hipStream_t);

#define DECL(c, s) extern "C" int zk_ntt_pass_c##c##_s##s(int, const NttPassArgs*, hipStream_t);
#define ALL(c) DECL(c, 3) DECL(c, 4) DECL(c, 5) DECL(c, 6) DECL(c, 7) DECL(c, 8) DECL(c, 9)
ALL(0)
ALL(1)

extern "C" NttPassLauncher zk_ntt_pass_launcher(int curve, int s) {
    static const NttPassLauncher tab[2][7] = {
        {zk_ntt_pass_c0_s3, zk_ntt_pass_c0_s4, zk_ntt_pass_c0_s5, zk_ntt_pass_c0_s6, zk_ntt_pass_c0_s7, zk_ntt_pass_c0_s8,
         zk_ntt_pass_c0_s9},
        {zk_ntt_pass_c1_s3, zk_ntt_pass_c1_s4, zk_ntt_pass_c1_s5, zk_ntt_pass_c1_s6, zk_ntt_pass_c1_s7, zk_ntt_pass_c1_s8,
         zk_ntt_pass_c1_s9},
    };
    if (curve < 0 || curve > 1 || s < 3 || s > 9) return nullptr;
    return tab[curve][s - 3];
}
