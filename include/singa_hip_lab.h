/* Test- and lab-only switches of libsinga_hip.so.  NOT part of the drop-in C ABI (include/singa_hip.h): nothing under singa_amd/
 * model code calls these; tests/ and tools/lab/ use them to reach kernel variants the automatic choice would not take. */
#ifndef SINGA_HIP_LAB_H
#define SINGA_HIP_LAB_H
#ifdef __cplusplus
extern "C" {
#endif
/* force the 128 x 128 (0) or the 64 x 64 (3) tile shape of singa_gemm_f32 wherever the automatic choice is between those two;
 * -1 = automatic */
int singa_gemm_force_cfg(int cfg);
/* resident workgroups per CU of one GEMM kernel variant (cfg 0: 128x128, 1: 128x32, 2: 32x128, 3: 64x64 tiles) */
int singa_gemm_occupancy(int a_r_contig, int b_r_contig, int cfg);
/* 1 selects the VALU (lane-broadcast) form of singa_so3_skinny_expand / _reduce, 0 (default) the matrix-core form */
int singa_so3_skinny_variant(int valu);
/* singa_lap_pe: graphs whose largest connected component has at least n atoms take the sparse route (Chebyshev-filtered subspace
 * iteration); default 384, a value above 896 switches it off, a small one forces it (tests) */
int singa_lap_pe_fsi_min(int n);
#ifdef __cplusplus
}
#endif
#endif /* SINGA_HIP_LAB_H */
