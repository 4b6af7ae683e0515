// wgrad16c.h - internal interface of the filter-column weight-gradient kernel (wgrad16c.hip), used by the grouped call in igemm16.hip.
#pragma once
#include <hip/hip_runtime.h>

#include "../../include/ctgan_hip.h"

#define CTGAN_WC_MAXCOL 10

// one (x, dy) use of a filter: `out` = this problem's first slab ([splits][R*S*C (+1)][K] fp32), `chunk` = pixels per split
struct ctgan_wc_problem {
    const ctgan_conv_desc* d;
    const float* x; const float* dy; float* out;
    int N;                 // rows (samples) of this use
    int relu_x;            // x -> relu(x) while it is staged
    int with_bias;         // 0: no bias row; 1: slab row R*S*C receives the column sums of dy; 2: the row exists and stays zero
    int chunk;
};

bool ctgan_wgrad16c_takes(const ctgan_conv_desc* d, int mma, int max_rows);
int ctgan_wgrad16c_tiles(const ctgan_conv_desc* d, int mma);                          // workgroups per split
void ctgan_wgrad16c_plan(const ctgan_wc_problem* probs, int n, int mma, int* chunks); // pixels per split of every problem (multiples of 64)
int ctgan_wgrad16c_launch(const ctgan_wc_problem* probs, int n, int mma, hipStream_t st);
