/* Geometry of one weight-normalised MLP as the gfx950 kernels see it.
 *
 * Plain C, shared by the host side (filled through ctypes) and the kernels.
 * Activations live in "slots": 16-wide tiles whose order the host chooses once
 * (monosdf_amd/plan.py); a slot maps to an original feature index or to padding.
 * See DESIGN.md "Data layout".
 */
#ifndef MONOSDF_PLAN_H
#define MONOSDF_PLAN_H

#include <stdint.h>

#define MSDF_MAX_TILES 17  /* 17 * 16 = 272 slots: 256 hidden + one spare tile */
#define MSDF_MAX_LAYERS 10

#define MSDF_PRECISION_F32 0
#define MSDF_PRECISION_BF16X3 1
#define MSDF_PRECISION_BF16X6 2   /* three bf16 planes per operand, six products: fp32-grade */

typedef struct {
  int32_t kt;        /* input tiles  (16 slots each) */
  int32_t ot;        /* output tiles */
  int32_t wf_off;    /* float4 offset of the forward pack  [ot_even][kt][64 lanes] */
  int32_t wb_off;    /* float4 offset of the transposed pack [kt_even][ot][64 lanes] */
  int32_t bias_off;  /* float offset of the packed bias [ot*16] */
  int32_t skip_tile; /* >=0: input tiles [skip_tile, skip_tile+in0_tiles) hold the network input */
  int32_t hpre;      /* sum of 16*ot of the hidden layers before this one (slot prefix) */
  int32_t qpre;      /* sum of 16*kt of the layers before this one */
  int32_t abpre;     /* sum of 16*ot of all layers before this one (incl. non-hidden) */
  int32_t ktp;       /* K of the forward pack, padded with zero weights to one of {3,5,16,17} */
  int32_t otp;       /* K of the transposed pack, padded to one of {16,17} */
  int32_t pad_;
} msdf_layer_t;

typedef struct {
  int32_t n_layers;     /* linear layers, last one is the output layer */
  int32_t e_tiles;      /* tiles holding the positional encoding (3 for multires 6) */
  int32_t aux_tiles;    /* tiles of extra input features (hash grid: 2), after the PE tiles */
  int32_t n_freqs;      /* positional-encoding octaves */
  int32_t sdf_slot;     /* slot of the sdf value in the last layer's output (256) */
  int32_t feat_tiles;   /* tiles of the feature vector in the last layer's output (16) */
  int32_t hsum;         /* total hidden slots  = sum 16*ot over hidden layers */
  int32_t qsum;         /* total q-bar slots   = sum 16*kt over hidden layers */
  int32_t absum;        /* total a-bar slots   = sum 16*ot over all layers */
  int32_t wsdf_off;     /* float offset (packed bias buffer) of `out_rows` rows of the LAST layer's weight, each as
                           16 * kt floats in input-slot order: the sdf row (SDF network) / the three colour rows
                           (colour network) -- outputs the kernels form as dot products instead of a matrix product */
  int32_t mode;         /* colour network: 1 = idr input [x, PE(v), n, feat], 0 = nerf [PE(v), feat]; sdf: unused */
  int32_t out_act;      /* colour network: 0 sigmoid, 1 relu (if_hdr) */
  int32_t precision;    /* MSDF_PRECISION_*: which matrix core the kernels run this plan on.  BF16X3 / BF16X6 plans
                           count K in blocks of 32 slots (ktp / otp) and their wf_off / wb_off address bf16 plane packs
                           (2 / 3 planes of 1 KB per out tile and k block) */
  int32_t out_rows;     /* rows packed at wsdf_off: slots sdf_slot .. sdf_slot + out_rows - 1 of the last layer */
  msdf_layer_t layer[MSDF_MAX_LAYERS];
} msdf_plan_t;

/* one gather rule of the weight packer */
typedef struct {
  int32_t w_off;     /* float offset of this layer's [rows x cols] weight in the flat buffer */
  int32_t b_off;     /* float offset of this layer's bias in the flat bias buffer */
  int32_t rows, cols;
  int32_t rowmap_off; /* int offset: slot -> original row or -1, length 16*ot */
  int32_t colmap_off; /* int offset: slot -> original col or -1, length 16*kt */
  float scale;       /* 1/sqrt(2) on the skip layer, else 1 */
  int32_t pad_;
} msdf_packrule_t;

/* one weight-gradient work item: PART[split] = sum_{p in split} X[p][0:wx]^T Y[p][0:wy]
 * (+ optional column sums of X, + optional sum_p v[p] Y[p][:]).  X / Y / v are float offsets into one of the launch's
 * two operand buffers (msdf_wgrad: base0 = the saved activations, base1 = the feature tensor) -- the table holds no
 * device address, so it is built once per network and point count and never copied again (a table of absolute
 * addresses had to be re-sent, synchronously, whenever the allocator handed the step other blocks); the *_off fields
 * below them are float offsets into the partial buffer. */
typedef struct {
  int64_t x;                    /* float offset of X [P_pad, x_ld] in its buffer */
  int64_t y;                    /* float offset of Y [P_pad, y_ld] */
  int64_t v;                    /* float offset of v [P_pad] (bufs: no v when its byte is 0xff) */
  int64_t part_off;             /* [n_splits][wx*wy] */
  int64_t colsum_off;           /* [n_splits][wx] or < 0 */
  int64_t vrow_off;             /* [n_splits][wy] or < 0 */
  int32_t x_ld, y_ld;           /* row pitches of X and Y */
  int32_t wx, wy;               /* multiples of 16, <= 256; wy == 0: column sums only */
  int32_t n_splits;             /* this item's share of the points is cut into n_splits workgroups */
  int32_t bufs;                 /* buffer index (0 / 1) of x | y << 8 | v << 16 (0xff: none) */
} msdf_wgrad_item_t;

/* one reduction rule: dst[rowmap[i]*dst_ld + colmap[j]] = scale * sum_b PART[b][i*wy + j] */
typedef struct {
  int64_t part_off;   /* n_blocks contiguous blocks of wx*wy floats */
  int64_t dst_off;    /* float offset into the flat gradient buffer */
  int32_t n_blocks;
  int32_t wx, wy;     /* wy == 1 with colmap_off < 0: a vector indexed by rowmap (bias) */
  int32_t rowmap_off; /* int offset into maps (slot -> original row), or < 0: fixed row `fixed_row` */
  int32_t colmap_off; /* int offset into maps (slot -> original col), or < 0 */
  int32_t dst_ld;
  int32_t fixed_row;
  float scale;
} msdf_reduce_rule_t;

#endif
