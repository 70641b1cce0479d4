// hmx_tile_bf.hip -- the tile kernel k_tile (hmx_kernels.hip) built a second time with the split-bf16 form of its distance GEMM
// (v_mfma_f32_16x16x32_bf16 on three exact bf16 parts per fp32 operand, see tile_dots_bf_regs) and the three launchers that reach it:
// l_tile_static_bf, l_update_bf, l_chain_bf.  A translation unit of its own so that the two families compile side by side.
#define HMX_TILE_BF 1
#include "hmx_kernels.hip"
