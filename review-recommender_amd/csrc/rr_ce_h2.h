// rr_ce_h2.h -- K5, reference-precision mode on the fp16 matrix cores: the kernels of csrc/rr_ce_h2.hip as rr_ce.hip's host
// side launches them.  Not part of the C ABI.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

// "h2" = an fp32 matrix X[R][K] held as TWO fp16 planes, hi = fp16(x) and lo = fp16((x - hi) * 2048) (x = hi + lo / 2048 to
// 2^-22 |x|), in 16-byte UNITS of eight consecutive k, chunk-major: unit (plane p, chunk kc, row) at index
// ((p * K / 8 + kc) * row_stride + row).  `row_stride` >= R, a multiple of 256 for the token-side operand of ce_h2_gemm.
#define CE_H2_SCALE 2048.0f
#define CE_H2_INV_SCALE 4.8828125e-4f

#define CE_H2_EPI_F32 0        // out32[token][N] = x W^T + b                         (fp32, row-major)
#define CE_H2_EPI_H2 1         // out2 = h2(x W^T + b)
#define CE_H2_EPI_GELU_H2 2    // out2 = h2(gelu(x W^T + b)), gelu = x Phi(x) (the erf form)

// rows [R][K] fp32 row-major (device) -> h2 planes
void ce_h2_pack(const float* d_src, int R, int K, void* d_dst, int64_t row_stride, hipStream_t st);
// out = X W^T + bias, X: h2 [M tokens][K] (row stride xs), W: h2 [N][K] (row stride N); N % 128 == 0, K % 32 == 0.
// out32 (EPI_F32) has leading dimension N; out2 (the other epilogues) has N / 8 chunks and row stride `os`.
// *flag is OR-ed with 1 if a value that is split to fp16 falls outside its range (|v| > 65504).  CE_H2_EPI_H2 multiplies
// the features below `qcols` by `qscale` (the attention's 1 / sqrt(d) log2 e, folded into the query projection).
void ce_h2_gemm(int epi, const void* W2, int N, const void* X2, int64_t xs, int M, int K, const float* bias, float* out32,
                void* out2, int64_t os, unsigned* flag, hipStream_t st, float qscale = 1.f, int qcols = 0);
// softmax(Q K^T / sqrt(32)) V per (sequence, head): qkv = ONE h2 image of [T][1152] (Q columns pre-scaled by
// 1 / sqrt(32) * log2 e, then K, then V; row stride xs), ctx2 = the context as an h2 image (row stride os).
// cls_only: only query 0 of every sequence, written to row `sequence` of ctx2.
void ce_h2_attention(const void* qkv, int64_t xs, const int32_t* cu, int n_seqs, int max_len, void* ctx2, int64_t os, unsigned* flag,
                     int cls_only, hipStream_t st);
int ce_h2_set_attributes();
// word + position + type embeddings -> LayerNorm -> h32 (fp32 rows) + hx (h2, row stride xs)
void ce_h2_embed_ln(const int32_t* tok, const int32_t* typ, const int32_t* pos, int T, int vocab, int n_pos, int n_typ,
                    const float* we, const float* pe, const float* te, const float* g, const float* b, float eps, float* h32,
                    void* hx, int64_t xs, unsigned* flag, hipStream_t st);
// h32 = LayerNorm(y + h32) * g + b, and its h2 image
void ce_h2_add_ln(const float* y, float* h32, int T, const float* g, const float* b, float eps, void* hx, int64_t xs,
                  unsigned* flag, hipStream_t st);
