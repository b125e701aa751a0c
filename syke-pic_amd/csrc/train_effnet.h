// Launchers of train_effnet.hip (EfficientNet training kernels); see that file for what each one computes.
#pragma once
#include "spk_common.h"


int spk_launch_col_stats(const bf16_t* x, float* partials, int M, int C, int* blocks, hipStream_t s);
int spk_launch_bna_finalize(const float* partials, int count, int C, int c_log, double M, const float* gamma,
                            const float* beta, float* rmean, float* rvar, float* st, float eps, float momentum,
                            float* tmp, hipStream_t s);
int spk_launch_bna_apply(const bf16_t* raw, const float* scale, const float* shift, const bf16_t* res,
                         const float* rowscale, bf16_t* out, int M, int C, int HW, int act, hipStream_t s);
int spk_launch_bna_bwd_reduce(const bf16_t* g, const bf16_t* raw, const float* scale, const float* shift,
                              const float* mean, const float* invstd, const float* rowscale, float* partials, int M,
                              int C, int HW, int act, int* blocks, hipStream_t s);
int spk_launch_bna_bwd_finalize(const float* partials, int count, int C, int c_log, double M, const float* gamma,
                                const float* invstd, float* dgamma, float* dbeta, float* coef, float* tmp, hipStream_t s);
int spk_launch_bna_bwd_apply(const bf16_t* g, const bf16_t* raw, const float* scale, const float* shift,
                             const float* mean, const float* invstd, const float* coef, const float* rowscale,
                             bf16_t* dy, bf16_t* g_res, int res_accumulate, int M, int C, int HW, int act,
                             hipStream_t s);
int spk_launch_sd_rowscale(float* rs, int n, float p, unsigned long long seed, hipStream_t s);
int spk_launch_stem3_train_fwd(const bf16_t* x, const float* w, bf16_t* y, int n, int h, int wd, int wstride, int cin,
                               int cout, int C, int ho, int wo, hipStream_t s, float out_scale = 1.0f);
int spk_stem3_wgrad_blocks(int M, int* pix_per_block);
int spk_launch_stem3_wgrad(const bf16_t* x, const bf16_t* dy, float* partials, int n, int h, int wd, int wstride,
                           int cin, int cout, int C, int ho, int wo, int* blocks, hipStream_t s);
int spk_launch_dw_train_fwd(const bf16_t* x, const float* wt, bf16_t* y, int n, int h, int wd, int C, int k, int stride,
                            int pad, int ho, int wo, hipStream_t s);
int spk_launch_dw_dgrad(const bf16_t* dy, const float* wt, bf16_t* dx, int accumulate, int n, int h, int wd, int C, int k,
                        int stride, int pad, int ho, int wo, hipStream_t s);
int spk_dw_wgrad_rows(int M, int C);
int spk_launch_dw_wgrad(const bf16_t* x, const bf16_t* dy, float* partials, int n, int h, int wd, int C, int c_log,
                        int k, int stride, int pad, int ho, int wo, int* rows, hipStream_t s);
int spk_launch_bna_apply_pool(const bf16_t* raw, const float* scale, const float* shift, bf16_t* out, float* part, int n,
                              int HW, int C, int act, hipStream_t s);
int spk_se_chunks(int HW);
int spk_launch_pool_rows(const bf16_t* x, const bf16_t* y, float* part, int n, int HW, int C, hipStream_t s);
int spk_launch_se_scale(const bf16_t* a, const float* gate, bf16_t* out, int n, int HW, int C, hipStream_t s);
int spk_launch_se_bwd_apply(const bf16_t* g, const float* gate, const float* dpool, bf16_t* da, int n, int HW, int C,
                            hipStream_t s);
int spk_launch_slab_reduce_sub(const float* slabs, float* out, int cout, int taps, int cin, int cout_p, int cin_p,
                               int splits, hipStream_t s);
// Every weight image of an EfficientNet training step in a few launches: entries of kind 0 / 1 are the forward / data-gradient
// GEMM images of a conv (bf16, channel-padded, offsets in elements into pbuf / wpack), kind 2 the tap-major depthwise
// window and its flipped copy (floats, offset into dwt; cin unused, cout_p = the padded channel count)
struct PadPackEntry {
  size_t src, dst;
  unsigned cout, taps, cin, cout_p, cin_p, kind;
};
struct PadPackTable {
  PadPackEntry e[48];
  int count;
};
int spk_launch_pack_padded_multi(const float* pbuf, bf16_t* wpack, float* dwt, const PadPackTable& t, hipStream_t s);
int spk_launch_se_gate_fwd(const float* part, int chunks, float scale, float* pooled, const float* W1, const float* b1,
                           const float* W2, const float* b2, float* u1, float* h1, float* gate, int n, int C, int Cl, int S,
                           hipStream_t s);
int spk_se_gate_tiles(int Cl, int S);
int spk_launch_se_gate_bwd(const float* pool_part, int chunks, float* dgate, const float* gate, const float* u1,
                           const float* W1, const float* W2, float* du1, float* dpool, float* part, int n, int C, int Cl,
                           int S, hipStream_t s);
int spk_launch_se_wgrad(const float* du2, const float* h1, const float* du1, const float* pooled, float* gW1, float* gb1,
                        float* gW2, float* gb2, int n, int C, int Cl, int S, hipStream_t s);
