// Packed weight image of the standard pair encoder (F = 40, ppffcs = [84, 32, 32, 16], train.py:35), shared by the
// forward kernels (pair_mlp.hip) and the backward kernel (pair_mlp_bwd.hip).  ONE function, std_pack_elem(), says
// which parameter every float of the image holds; the host pack (cppf_pair_mlp_pack) and the device pack
// (cppf_pair_mlp_pack_device, used when the weights change every step) both evaluate it, so the two images are
// identical by construction.
//
// MFMA operand convention (v_mfma_f32_16x16x4_f32): lane l = (j = l & 15, g = l >> 4); the A operand of k-step s is
// one float per lane, A[row j][k(s,g)], with k(s,g) = 16*(s/4) + 4*g + s%4 for the hidden layers (khid).
#pragma once
#include <stdint.h>

#ifndef __HIPCC__
#define CPPF_HD
#else
#define CPPF_HD __host__ __device__
#endif

#define STD_F 40
#define STD_NOB 9                      // final layer padded to 9 x 16 = 144 outputs
#define STD_NOBP 12                    // floats per lane per k-step of the final layer (3 x b128)
#define OFF_W0P 0                       // [64][4]     PPF k-step of layer 0 (k = 80 + g): ob0,1 = fc1, ob2,3 = fc0
#define OFF_W0B (OFF_W0P + 64 * 4)      // [8][64][2]  layer 0 fc2
#define OFF_W1A (OFF_W0B + 8 * 64 * 2)  // [8][64][2]  layer 1 fc1
#define OFF_W1B (OFF_W1A + 8 * 64 * 2)  // [8][64][2]  layer 1 fc2
#define OFF_W2 (OFF_W1B + 8 * 64 * 2)   // [8][64][2]  ob0 = layer 2 fc1 (16), ob1 = layer 2 fc0 (16)
#define OFF_W2B (OFF_W2 + 8 * 64 * 2)   // [4][64][1]  layer 2 fc2
#define OFF_WF (OFF_W2B + 4 * 64)       // [4][64][12] final, 9 used
#define OFF_B0B (OFF_WF + 4 * 64 * STD_NOBP)  // biases in natural output order
#define OFF_B1A (OFF_B0B + 32)
#define OFF_B1B (OFF_B1A + 32)
#define OFF_B2 (OFF_B1B + 32)
#define OFF_B2B (OFF_B2 + 32)
#define OFF_BF (OFF_B2B + 16)
#define STD_LDS (OFF_BF + 144)          // 7 968 floats = 31 872 B live in LDS
#define OFF_WPT STD_LDS                 // [40][128]   per-point projection: column r < 64: W[r][k], r >= 64: W[r-64][40+k]
#define OFF_BPT (OFF_WPT + 40 * 128)    // [64]        fc1 | fc0 bias of layer 0 (folded into the feat_a table)
#define OFF_WFD (OFF_BPT + 64)          // [4][64][12] final layer with the output columns in DECODE order (dec_col)
#define OFF_BFD (OFF_WFD + 4 * 64 * STD_NOBP)  // [144] its bias, slot order
// ---- backward section: TRANSPOSED weights as A operands, A[input row 16*ib + j][k = output khid(s,g)] = W[k][16*ib + j]
#define OFF_T0B (OFF_BFD + 144)         // [8][64][2]  layer 0 fc2^T
#define OFF_T1A (OFF_T0B + 8 * 64 * 2)  // [8][64][2]  layer 1 fc1^T
#define OFF_T1B (OFF_T1A + 8 * 64 * 2)  // [8][64][2]  layer 1 fc2^T
#define OFF_T2 (OFF_T1B + 8 * 64 * 2)   // [4][64][4]  layer 2: fc0^T ib0, fc0^T ib1, fc1^T ib0, fc1^T ib1  (32 rows, k over 16)
#define OFF_T2B (OFF_T2 + 4 * 64 * 4)   // [4][64][1]  layer 2 fc2^T
#define OFF_TF (OFF_T2B + 4 * 64)       // [36][64]    final^T: rows = its 16 inputs, k over the 144 padded outputs
#define STD_BWD_FLOATS (OFF_TF + 36 * 64 - OFF_T0B)   // 6 656
#define STD_PACKED (OFF_TF + 36 * 64)   // 23 024 floats
#define PROJ_COLS 128

static CPPF_HD inline int khid(int s, int g) { return 16 * (s / 4) + 4 * g + (s % 4); }

// DECODE order of the 141 output columns (train.py:68-75: 2 x 32 centre bins, 2 x 36 angle bins, 2 sign logits, 3
// log-scales): accumulator slot (block ob, lane group g, register r) holds
//   ob 0..7 : bin 8g (or 9g) + 4*(ob & 1) + r of head ob / 2 -- a lane owns a run of consecutive bins of every head
//   ob 8    : r = 0, 1: the 9th bin (9g + 8) of the up / right head;  r = 2, 3: aux_up aux_right | sx sy | sz - | - -  for g = 0..3
// so that the in-register sampler needs two lane exchanges per statistic and no per-row bookkeeping.
static CPPF_HD inline int dec_col(int ob, int g, int r)
{
    if (ob < 4) return 32 * (ob >> 1) + 8 * g + 4 * (ob & 1) + r;
    if (ob < 8) return 64 + 36 * ((ob - 4) >> 1) + 9 * g + 4 * (ob & 1) + r;
    if (r < 2) return 64 + 36 * r + 9 * g + 8;
    const int q = 2 * g + (r - 2);   // 0..7 -> columns 136..140, then unused
    return q < 5 ? 136 + q : -1;
}

// Float `idx` of the packed image.  params: flat fp32 buffer of torch-layout tensors; offs: 6 per res layer {fc1.w, fc1.b,
// fc2.w, fc2.b, fc0.w | -1, fc0.b | -1}, then final.w, final.b (layers 0 and 2 have fc0, layer 1 does not).
static CPPF_HD inline float std_pack_elem(int idx, const float* params, const int64_t* offs, int out_dim)
{
    const float *w1_0 = params + offs[0], *b1_0 = params + offs[1], *w2_0 = params + offs[2], *b2_0 = params + offs[3];
    const float *w0_0 = params + offs[4], *b0_0 = params + offs[5];
    const float *w1_1 = params + offs[6], *b1_1 = params + offs[7], *w2_1 = params + offs[8], *b2_1 = params + offs[9];
    const float *w1_2 = params + offs[12], *b1_2 = params + offs[13], *w2_2 = params + offs[14], *b2_2 = params + offs[15];
    const float *w0_2 = params + offs[16], *b0_2 = params + offs[17];
    const float *wf = params + offs[18], *bf = params + offs[19];
    if (idx < OFF_W0B) {                       // [64][4]
        const int l = idx >> 2, ob = idx & 3;
        const int o = 16 * (ob & 1) + (l & 15), k = 80 + (l >> 4);
        return (ob < 2 ? w1_0 : w0_0)[o * 84 + k];
    }
    if (idx < OFF_W2) {                        // three [8][64][2] blocks of 32 -> 32 layers
        const int which = (idx - OFF_W0B) / 1024, i = (idx - OFF_W0B) % 1024;
        const int s = i >> 7, l = (i >> 1) & 63, ob = i & 1;
        const float* w = which == 0 ? w2_0 : (which == 1 ? w1_1 : w2_1);
        return w[(16 * ob + (l & 15)) * 32 + khid(s, l >> 4)];
    }
    if (idx < OFF_W2B) {                       // [8][64][2]: fc1 | fc0 of layer 2 (32 -> 16)
        const int i = idx - OFF_W2, s = i >> 7, l = (i >> 1) & 63, ob = i & 1;
        return (ob == 0 ? w1_2 : w0_2)[(l & 15) * 32 + khid(s, l >> 4)];
    }
    if (idx < OFF_WF) {                        // [4][64]
        const int i = idx - OFF_W2B, s = i >> 6, l = i & 63;
        return w2_2[(l & 15) * 16 + khid(s, l >> 4)];
    }
    if (idx < OFF_B0B) {                       // [4][64][12]
        const int i = idx - OFF_WF, s = i / (64 * STD_NOBP), l = (i / STD_NOBP) & 63, ob = i % STD_NOBP;
        const int o = 16 * ob + (l & 15);
        return (ob < STD_NOB && o < out_dim) ? wf[o * 16 + khid(s, l >> 4)] : 0.f;
    }
    if (idx < OFF_B1A) return b2_0[idx - OFF_B0B];
    if (idx < OFF_B1B) return b1_1[idx - OFF_B1A];
    if (idx < OFF_B2) return b2_1[idx - OFF_B1B];
    if (idx < OFF_B2B) { const int o = idx - OFF_B2; return o < 16 ? b1_2[o] : b0_2[o - 16]; }
    if (idx < OFF_BF) return b2_2[idx - OFF_B2B];
    if (idx < OFF_WPT) { const int o = idx - OFF_BF; return o < out_dim ? bf[o] : 0.f; }
    if (idx < OFF_BPT) {                       // [40][128]
        const int i = idx - OFF_WPT, k = i >> 7, r = i & 127, oc = r & 63;
        const float* w = oc < 32 ? w1_0 + oc * 84 : w0_0 + (oc - 32) * 84;
        return w[(r < 64 ? 0 : 40) + k];
    }
    if (idx < OFF_WFD) { const int o = idx - OFF_BPT; return o < 32 ? b1_0[o] : b0_0[o - 32]; }
    if (idx < OFF_BFD) {                       // [4][64][12], decode column order
        if (out_dim != 141) return 0.f;
        const int i = idx - OFF_WFD, s = i / (64 * STD_NOBP), l = (i / STD_NOBP) & 63, ob = i % STD_NOBP;
        if (ob >= STD_NOB) return 0.f;
        const int m = l & 15, c = dec_col(ob, m >> 2, m & 3);
        return c >= 0 ? wf[c * 16 + khid(s, l >> 4)] : 0.f;
    }
    if (idx < OFF_T0B) {
        if (out_dim != 141) return 0.f;
        const int i = idx - OFF_BFD, c = dec_col(i >> 4, (i & 15) >> 2, i & 3);
        return c >= 0 ? bf[c] : 0.f;
    }
    if (idx < OFF_T2) {                        // three [8][64][2] transposed 32 x 32 blocks
        const int which = (idx - OFF_T0B) / 1024, i = (idx - OFF_T0B) % 1024;
        const int s = i >> 7, l = (i >> 1) & 63, ib = i & 1;
        const float* w = which == 0 ? w2_0 : (which == 1 ? w1_1 : w2_1);
        return w[khid(s, l >> 4) * 32 + 16 * ib + (l & 15)];
    }
    if (idx < OFF_T2B) {                       // [4][64][4]: layer 2 fc0^T (ib 0,1), fc1^T (ib 0,1); W is [16][32]
        const int i = idx - OFF_T2, s = i >> 8, l = (i >> 2) & 63, q = i & 3;
        return (q < 2 ? w0_2 : w1_2)[khid(s, l >> 4) * 32 + 16 * (q & 1) + (l & 15)];
    }
    if (idx < OFF_TF) {                        // [4][64]
        const int i = idx - OFF_T2B, s = i >> 6, l = i & 63;
        return w2_2[khid(s, l >> 4) * 16 + (l & 15)];
    }
    {                                          // [36][64]
        const int i = idx - OFF_TF, s = i >> 6, l = i & 63, k = khid(s, l >> 4);
        return k < out_dim ? wf[k * 16 + (l & 15)] : 0.f;
    }
}
