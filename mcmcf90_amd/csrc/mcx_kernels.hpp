// mcx_kernels.hpp -- HIP kernels of the adaptive-Metropolis engine (gfx950, wave64): the umbrella over the family headers.
//
// Execution model: one lane = one chain, one 64-lane wave (= one workgroup) = one "tile" of 64 chains.  All per-chain arrays in HBM
// are tile-interleaved,
//     element k of chain (tile, lane)  ->  base[(tile*K + k)*64 + lane],
// i.e. parameter-major inside a tile, so every wave access is one contiguous 512-byte segment and a tile's whole Cholesky factor is
// one sequential stream.  The O(d^2) sweeps over a chain's packed factor are left-looking COLUMN PANELS: PW = 8 columns of per-lane
// state (rotation work values, proposal accumulators) live in registers with compile-time indices, every row contributes one
// contiguous PW x 512-byte segment (streamed non-temporally), and the few O(d) vectors (normals, rotations, candidate) sit in
// per-chain global scratch, with the most re-read rotations cached in LDS.  Nothing here is templated on npar.
//
//   mcx_common.hpp    EngineDev, layout macros, device targets, the normal generator
//   mcx_products.hpp  R'z on the packed triangle (column panels), full-matrix products, the lane SVD, shared-table forms
//   mcx_step.hpp      step_kernel<RAM,DR,POOLED> and its forms (the headline kernel: step_kernel_ram_wide), ram_update, dr_body
//   mcx_scam.hpp      scam_kernel / scam_mw_kernel (per-chain rotation), scam_pooled(12)_kernel (pooled rotation, f64 MFMA)
//   mcx_pooled.hpp    pooled_mfma_kernel (pooled AM / RAM / ER / DR on the f64 matrix cores)
//   mcx_pooled_ks.hpp pooled_mfma_ks_kernel (the same with a forty-row LDS vector in two pieces: eight tiles per CU at npar 41..64)
//   mcx_phase.hpp     host_phase_kernel<0..7>, dev_eval_kernel, step_kernel_cols (nycol >= 1), run1_kernel
//   mcx_adapt.hpp     init_kernel, adapt_pre / adapt_cov_diag / adapt_cov_off / adapt_covb_* / adapt_post kernels
//   mcx_svd.hpp       svd_sweep_stream(32)_kernel, svd_applyv_stream32_kernel, tile <-> chain layout conversion
//   mcx_moments.hpp   moments_kernel, moments_tree_kernel, debug kernels
// The lane-GROUP kernels (16 / 4 lanes per chain, factors on chip) are mcx_group.hpp and mcx_group_ram.hpp.
// Kernel forms that were measured and lost, or that a later form superseded, are not in the product library: tools/variants/README.md.
#pragma once
#include "mcx_common.hpp"
#include "mcx_products.hpp"
#include "mcx_step.hpp"
#include "mcx_scam.hpp"
#include "mcx_pooled.hpp"
#include "mcx_pooled_ks.hpp"
#include "mcx_phase.hpp"
#include "mcx_adapt.hpp"
#include "mcx_svd.hpp"
#include "mcx_moments.hpp"
