// MARS encoder tail (conv4_x): crop-resident, weight-stationary kernels of csrc/mars_tail.hip, launched by the executor in
// csrc/nets.hip.  Architecture: tools/freeze_model.py:13-86,118-156 (upstream paths).
#pragma once
#include "common.h"

struct MarsWsP {
    const _Float16 *in;  int cs_in, coff_in;        // stride 1: [n][8][4][128]; stride 2: [n][16][8][64] (NHWC f16, channel stride cs)
    const _Float16 *in2; int cs_in2, coff_in2;      // stride 2: the tensor the 1x1 stride-2 projection reads (the block's raw input)
    const _Float16 *w;   int kpad;                  // [128][kpad] f16, k = tap * Cin + channel
    const _Float16 *w2;  int kpad2;                 // projection [128][kpad2]
    const float *bias, *bias2;                      // [128] each
    const _Float16 *res; int cs_res, coff_res;      // stride 1: residual input (mode 1)
    _Float16 *out;  int cs_out, coff_out;
    _Float16 *out2; int cs_out2, coff_out2;         // mode 1: ELU(scale * v + shift) view (aff2 = [2][cout_pad] f32); mode 2: the projection's output
    const float *aff2; int cout_pad;
    const _Float16 *zero;                           // >= 16 zero bytes
    int n_img;
};

enum { MARS_WS_S1 = 0, MARS_WS_S1_RES = 1, MARS_WS_S2_PROJ = 2 };

// mode: MARS_WS_*; act: ACT_* of nets.hip (0 none, 2 ELU) applied to the main output before the residual; out2: mode 1 only
int mars_ws128_launch(hipStream_t s, int device, const MarsWsP &P, int mode, int act, bool out2);

// ---- conv3_x: two layers per launch (csrc/mars_pair.hip)
struct MarsPairP {
    const _Float16 *in;  int cs_in, coff_in;        // first = true: the block's pre-activation [n][31][15][32]; else [n][16][8][64]
    const _Float16 *in2; int cs_in2, coff_in2;      // first = true: the raw block input (what the 1x1 stride-2 projection reads)
    const _Float16 *wa;  int kpad_a; const float *bias_a;       // stage A: 3x3 (stride 2, 32 -> 64 | stride 1, 64 -> 64) + ELU
    const _Float16 *wp;  int kpad_p; const float *bias_p;       // first = true: the projection [64][kpad_p]
    const _Float16 *wb;  int kpad_b; const float *bias_b;       // stage B: 3x3 64 -> 64, no activation, + residual
    const _Float16 *res; int cs_res, coff_res;      // first = false: the residual tensor [n][16][8][64] (first: the projection, kept on chip)
    _Float16 *out;  int cs_out, coff_out;
    _Float16 *out2; int cs_out2, coff_out2;         // ELU(scale * v + shift): aff2 = [2][cout_pad] f32
    const float *aff2; int cout_pad;
    const _Float16 *zero;
    int n_img;
};
int mars_pair64_launch(hipStream_t s, int device, const MarsPairP &P, bool first);
