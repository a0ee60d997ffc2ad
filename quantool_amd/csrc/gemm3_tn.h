// fp32-accurate "TN" product on the bf16 MFMA:  acc[m][n] = sum_k A[k][m] * B[k][n]  with every fp32
// operand split into three bf16 planes  x = hi + mid + lo  (each the round-to-nearest bf16 of what the
// previous ones left: the residual after three planes is <= 2^-27 |x|) and the six plane products whose
// weight is >= 2^-16 of hi*hi summed in one fp32 MFMA accumulator, smallest first:
//     lo*hi, hi*lo, mid*mid, mid*hi, hi*mid, hi*hi              (dropped: mid*lo, lo*mid, lo*lo <= 2^-24)
// -- 6 bf16-MFMA flops per fp32 flop at 16x the f32-MFMA rate.  The kernel is xtx.hip's LDS-ring pipeline
// (ring_pipe.h) with the A and B panels of a unit taken from two plane sets; it serves the two products
// that carry the K^3 of qt_cholesky_inverse_upper (cholesky.hip), which are not bit-pinned by the oracle
// (the factor U is compared through the sweep's outputs, not bit for bit).  NOT used by the
// GPTQ sweep's trailing update, whose ascending-k fmaf order is part of the parity contract (sgemm_tn.h).
#pragma once
#include <vector>

#include "common.h"

constexpr int G3_CHUNK_ROWS = 64;   // granularity of an item's k range

// One workgroup = one item: a 256x256 output tile over k rows [64*c_lo, 64*c_hi) of the planes (>= 2 chunks).
struct G3Item {
    int tile;   // (ti << 16) | tj : output rows 256*ti.., columns 256*tj..
    int c_lo;   // k range in chunks of G3_CHUNK_ROWS rows, relative to row0 of the call
    int c_hi;
    int slab;   // < 0: the item covers the tile's whole k range and applies it to C itself (C -= acc / C = acc);
                // >= 0: partial product -> slab[slab] (256x256 fp32), reduced in table order afterwards
};
// One output tile assembled from `count` consecutive slabs starting at `first`.
struct G3Red {
    int tile;
    int first;
    int count;
    int pad;
};

enum { G3_SUB = 0, G3_SET = 1, G3_ADD = 2 };   // C -= product  /  C = product  /  C += product

struct G3Args {
    const unsigned short* Apl;   // plane 0 of the A operand, [rows][ld] bf16; plane q at + q * plane_stride
    const unsigned short* Bpl;
    int64_t plane_stride;        // elements between planes (same for A and B)
    int64_t ld;                  // row pitch in elements (same for A and B; a multiple of 8)
    int64_t rowA0, rowB0;        // plane row of k = 0
    int colA0, colB0;            // plane column of output row 0 / output column 0
    int colmax;                  // planes have colmax columns (loads are clamped to colmax - 8)
    int M, N;                    // output extent
    float* C;
    int64_t ldc;
    // B-side layout when it differs from A's (0 = same as A): pitch, plane stride, column count
    int64_t ldB = 0, plane_strideB = 0;
    int colmaxB = 0;
    int mode;                    // G3_SUB / G3_SET: what direct items and the slab reduction do to C
    float* slabs;                // [n_slabs][256*256]
    const G3Item* items;         // device
    int n_items;
    const G3Red* red;            // device
    int n_red;
    // batch > 1: `batch` problems of identical shape in one launch (one item / reduction table for all): problem b
    // reads planes at Apl + b * bsApl, Bpl + b * bsBpl (elements), writes C + b * bsC and uses slabs + b * bsSlabs.
    int batch = 1;
    int64_t bsApl = 0, bsBpl = 0, bsC = 0, bsSlabs = 0;
};

// Enqueue the product (and the slab reduction when n_red > 0).  Returns qt_status.
int qt_gemm3_launch(const G3Args& a, hipStream_t stream);

// fp32 [rows][ld_src] -> three bf16 planes at planes + q * plane_stride (pitch ld_pl).  cols % 4 == 0.
// mask_upper != 0: elements with (col_g0 + c) > (row_g0 + r) are written as zeros (a lower-triangular
// source whose upper part is not initialised).
// batch > 1: problem b reads src + b * bs_src and writes planes + b * bs_planes (elements).
int qt_split3_launch(const float* src, int64_t ld_src, int rows, int cols, unsigned short* planes, int64_t ld_pl,
                     int64_t plane_stride, int mask_upper, int row_g0, int col_g0, hipStream_t stream, int batch = 1,
                     int64_t bs_src = 0, int64_t bs_planes = 0);

// Host-side item table of a block-row product: Tm x Tn tiles over k chunks [lo, c_end) (chunks of
// G3_CHUNK_ROWS rows), lo = 0, or with tri != 0 lo = first chunk of row 256 * tj (B is lower-triangular in
// 256-blocks: B[k][n] == 0 for k < 256 * tj); split along k into <= 256 items of near-equal length (one round
// of the 256 CUs), longest first.
long g3_row_chunks(int Tm, int Tn, int c_end, int tri);
void g3_plan_row(int Tm, int Tn, int c_end, int tri, std::vector<G3Item>& items, std::vector<G3Red>& red, int target_items = 0);
