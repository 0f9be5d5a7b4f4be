// generated: balanced radix-2^30 digits of the BLS12-381 base-field modulus and -q^-1 mod 2^30
static __device__ __constant__ int FS_P_DUMMY = 0;
#define FS_NL 13
constexpr int FS_P[13] = {-21845, -402915328, 356515836, -352321620, -252304353, 55215067, 288093811, 316751073, -321428361, 517541167, -375082566, -91332614, 1704210};
constexpr unsigned FS_PINV = 1073545213u;
