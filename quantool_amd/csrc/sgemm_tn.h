// fp32 "TN" GEMM on the f32-input MFMA (v_mfma_f32_32x32x2_f32):
//     acc[m][n] = sum_{k ascending} A[k][m] * B[k][n]      (both operands k-major, row-major)
//     MODE_SUB: Cout = Cin - acc     MODE_SET: Cout = acc     MODE_NEG: Cout = -acc
// The accumulation is bit-for-bit an ascending-k fmaf chain starting from 0 (guide section 3,
// "FP32-input MFMA"), which is the order oracle/gptq_oracle.c fixes for upstream's
// "W[:, i2:] -= Err1 @ Hinv[i1:i2, i2:]".  Used by the GPTQ trailing update (a11) and by the
// blocked Cholesky / triangular inverse (a8), whose recurrences are all written in this form.
#pragma once
#include "common.h"

enum { SG_MODE_SUB = 0, SG_MODE_SET = 1, SG_MODE_NEG = 2 };
enum { SG_K_FULL = 0, SG_K_FROM_N0 = 1 };  // SG_K_FROM_N0: B[k][n] == 0 for k < n (skip them)
constexpr int SG_MAX_GROUPS = 16;          // row groups of one stacked product / problems of one batched sweep
constexpr int SG_MAX_BATCH = 16;           // problems of one batched factorisation

struct SgemmArgs {
    const float* A; int64_t lda;
    const float* B; int64_t ldb;
    const float* Cin; int64_t ldcin;
    float* Cout; int64_t ldcout;
    int M, N, kdim;
    int k_mode;
    int mode;
    // split-K (optional): when split_ws != nullptr and the shape is latency-bound (few tiles,
    // long k), partial products go to slabs in split_ws and an ordered reduction applies `mode`.
    // Never used when bit-exact k order matters (the caller passes nullptr there).
    float* split_ws = nullptr;
    size_t split_ws_bytes = 0;
    // MODE_SUB only: when > 0 (a multiple of 16) the k range is a sequence of chains of this length
    // and Cout = (((Cin - chain_0) - chain_1) - ...), every chain an ascending-k fmaf chain from 0 --
    // bit for bit what chain-many separate launches would leave in C, with C read and written once
    // (the GPTQ sweep's far update over several 128-column blocks).
    int chain_len = 0;
    // M == N products of which only the upper triangle is wanted (the right-looking Cholesky update
    // A[J1:, J1:] -= R^T R): tiles entirely below the diagonal are skipped.  Cin == Cout (in place) is
    // allowed in every mode: a thread reads its C elements before it writes them and no other does.
    int upper_only = 0;
    // ---- several problems in one launch (results per problem are bit-identical to separate launches:
    // tile shape, split-K decision and k order are taken from ONE problem's shape, never from the batch) ----
    // batch > 1: `batch` independent problems of identical shape; problem b reads / writes operand + b * stride
    // (elements).  The Cholesky chains of a layer's equal-K Hessians (qt_cholesky_inverse_upper_batched).
    int batch = 1;
    int64_t bsA = 0, bsB = 0, bsCin = 0, bsCout = 0;
    int64_t bs_split = 0;          // split-K slabs of problem b at split_ws + b * bs_split (>= splits * M * N)
    // n_groups > 0: ONE product over stacked rows whose B operand depends on the row range: rows
    // [group_m_end[g-1], group_m_end[g]) of the output (columns of A) multiply B + g * group_bsB.  Boundaries are
    // multiples of 128 (no tile straddles two groups).  The sweeps of Linears with different factors U, stacked
    // (qt_gptq_sweep_grouped).
    int n_groups = 0;
    int64_t group_bsB = 0;
    int group_m_end[SG_MAX_GROUPS] = {};
    // internal (set by qt_sgemm_tn)
    int k_chunk = 0;
    int n_splits = 1;
    int fast_interior = 1;
};

// Enqueue on `stream`; picks the tile size from the problem shape.  Returns qt_status.
int qt_sgemm_tn(const SgemmArgs& a, hipStream_t stream);

// Ordered slab reduction (shared by the split-K paths): Cout = mode(Cin, sum_z slabs[z]).
int qt_splitk_reduce(const float* slabs, int splits, int M, int N, const float* Cin, int64_t ldcin, float* Cout,
                     int64_t ldcout, int mode, hipStream_t stream, int batch = 1, int64_t bs_slabs = 0, int64_t bs_cin = 0,
                     int64_t bs_cout = 0);
