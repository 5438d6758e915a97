// pr_tuning.h -- the compile-time knobs of the kernels, in one place.  Each is a NUMBER that was swept on hardware (profiles/NOTES.md holds the
// sweeps); -DPR_X=... on the hipcc line (PR_EXTRA_FLAGS of pose_refine_amd/build.py) overrides one for an A/B build, and tools/ab_libs.sh runs
// variant libraries side by side on one box.  The on/off switches of rounds 1-3 whose A/B is decided are gone: the winning branch is the code.
#pragma once

// ---- asynchronous path (pr_refine.cpp) ------------------------------------------------------------------------------------------------
#ifndef PR_SLOTS
#define PR_SLOTS 2                                              // asynchronous slots per context (pr_refine_submit's slot argument: 0 .. PR_SLOTS - 1)
#endif

// ---- correspondence pass (icp_pass.hip) --------------------------------------------------------------------------------------------
#ifndef PR_PASS_WAVES
#define PR_PASS_WAVES 1                                         // __launch_bounds__ minimum waves per SIMD of icp_pass_kernel: 4, 5 waves run within 0.2 % of the compiler's own choice (round 6: 277.8 / 277.1 k against 278.0 / 277.5 k); 6 spills the accumulators: 162 k
#endif
#ifndef PR_PASS_PREFETCH
#define PR_PASS_PREFETCH 0                                      // 1: the next 1024-point step's cloud points are loaded before this step's gathers are consumed (12 more VGPRs: 106, four waves per SIMD).  Same box: 262 k against 274 k poses/s -- not kept
#endif
// (Round 6, measured and not kept: the cloud points of a workgroup's steps 1 and 2 requested up front straight into LDS -- global_load_lds_dwordx3, 24 KB per workgroup, no registers in
// flight -- so that a workgroup's chain loses two of its three dependent trips for cloud points: 103 VGPRs / four waves per SIMD, or 96 with a spill at five; a lone 256-hypothesis launch
// 43.1 / 43.4 against 38.5 us, pipelined 256-258 k / 263-265 k against 275 k poses/s, same box, three rounds.  The LDS-DMA path lands 768-byte pieces slower than plain loads return.)
// (Round 6, what the cloud's write-back costs the projective pass and what deferring it returns -- measured, not kept.  TIMING ONLY, results wrong: the pass without its
// stores runs 32.5 against 38.4 us per 256-hypothesis launch and the pipeline 315 k against 277 k poses/s -- the 12 bytes per point written back are a sixth of the launch.
// Bit-identical form: the updates of a hypothesis kept in a ring of K records, the cloud stored on every K-th pass only, the passes between applying the updates since
// the last store one after the other in registers (the reference's own operations, so every point is the same float): K = 2 / 3 / 4 / 5 -> 268-283 / 279-281 / 274 / 269 k
// against 274-276 k for K = 1 in the same build and 277-279 k for the present code -- the chain's extra transforms (14 issue slots each) and scalar loads give back what
// the stores saved: the kernel behaves as the SUM of its parts (a wavefront issues a VALU instruction every 4.75 cycles at best and two are rarely ready at once).
// More wavefronts by force: 6 per SIMD with two gathers in flight per lane (8 spilled registers) 256-268 k, one gather 235-248 k, 7 per SIMD 217-241 k; two gathers at
// five wavefronts 270-276 k.  Five wavefronts, four gathers, every pass writing back stays.)
#ifndef PR_PASS_LATE_STORE
#define PR_PASS_LATE_STORE 1                                     // projective pass: a step's write-back is issued BEHIND its four gathers, not ahead of them (the stores no longer sit in front of the gathers in the wavefront's memory queue).  Round 6, same box, six alternating 100-step runs: 280.1 / 280.2 / 278.4 / 281.0 / 281.1 / 280.3 k against 275.2 / 276.7 / 276.0 / 278.5 / 278.2 / 278.1 k poses/s; behind the step's accumulation instead: 278.5-279.9 k; as non-temporal stores: 279.1-280.6 k
#endif
#ifndef PR_HOST_ROW_TAG
#define PR_HOST_ROW_TAG 1                                        // PR_SOLVE_HOST with group flags: every row also carries the iteration's tag behind its sums, checked by the host (a flag that overtook a row costs a stream wait, never a wrong solve)
#endif
#ifndef PR_PASS_TAIL_PRIO
#define PR_PASS_TAIL_PRIO 0                                     // s_setprio of a wavefront that has entered the reduction tail of icp_pass_kernel (0: none).  Round 6, same box, 100 steps x 2: priority 3 278.3 / 278.3 k against 278.0 / 277.5 k poses/s -- inside the noise, not kept
#endif
#ifndef PR_GATHER_BATCH
#define PR_GATHER_BATCH 4                                       // projective scene gathers issued back to back before the first is tested (all four points of a lane)
#endif

// ---- raster (raster.hip) ----------------------------------------------------------------------------------------------------------
#ifndef PR_RASTER_CHUNKS
#define PR_RASTER_CHUNKS 1                                      // fused path: the raster of a (sub-)batch as this many launches over consecutive hypotheses.  Same box, 100 steps: 1 -> 274.1 k, 2 -> 271.7 k, 4 -> 271.1 k poses/s -- not kept
#endif

// ---- render -> cloud (d2c.hip) ----------------------------------------------------------------------------------------------------
#ifndef PR_CLOUD_PACK
#define PR_CLOUD_PACK 32                                        // fused path: a hypothesis' cloud starts where the one before it ends, rounded up to this many points (32 = 128-byte lines; 0: fixed stride of the largest pixel box, rounds 1-4)
#endif

#ifndef PR_BOX_PACK
#define PR_BOX_PACK 32                                          // fused asynchronous path: a hypothesis' pixel box is its own little image in the depth workspace, packed behind the one before it, rounded up to this many pixels (0: full frames, rounds 1-4)
#endif

// ---- kd-tree search: ordered per-lane walks (nn_query.h) ---------------------------------------------------------------------------
#ifndef PR_LEAF_BATCH
#define PR_LEAF_BATCH 10                                        // points of a leaf fetched before the first compare (the reference's max_leaf)
#endif
#ifndef PR_NN_WIDE_BOUND
#define PR_NN_WIDE_BOUND 4.0e-6f                                // (2 mm)^2: above this bound a node's whole 32-byte record is fetched at once, below it the 8-byte descent word first
#endif

// ---- kd-tree search: pixel grid (nn_query.h, nn_search.hip) ------------------------------------------------------------------------
#ifndef PR_GRID_MAXW
#define PR_GRID_MAXW 2                                          // half-width of the largest pixel window scanned (5 x 5 cells); a larger window goes to the tree
#endif
#ifndef PR_RING_ROWS
#define PR_RING_ROWS 2                                          // rows of a 6 x 6 ring of the descent in flight at a time (1, 2, 3, 6 measured: not latency-bound)
#endif
#ifndef PR_DESCENT_LATE
#define PR_DESCENT_LATE 1                                       // the descent's first ring from pass 1 on: 0 = as in pass 0 (5 x 5 blocks of 16 x 16 pixels), 1 = 3 x 3 such blocks, 2 = none (straight to 5 x 5 blocks of 4 x 4 pixels)
#endif
#ifndef PR_RING16_W
#define PR_RING16_W 5                                           // cells per side of the first ring (16 x 16-pixel blocks around the query's own projection)
#endif
#ifndef PR_RING_STAGED
#define PR_RING_STAGED 1                                        // 1: a ring's rows are loaded PR_RING_ROWS at a time (72 VGPRs: seven wavefronts per SIMD); 0: all at once (101 VGPRs: four)
#endif
#ifndef PR_BOUND_WAVES
#define PR_BOUND_WAVES 1                                        // __launch_bounds__ minimum waves per SIMD of nn_bound_kernel (1: the compiler's own choice)
#endif
#ifndef PR_RING_W
#define PR_RING_W 5                                             // cells per side of a ring of the descent: a block's 4 x 4 children and the row / column BEFORE them (PR_RING_OFF = 1).
#endif                                                          // Same box, configs[2] pipelined: 6 / off 1 (both neighbours, rounds 2-4) 41.8 k poses/s, 5 / 1 43.0 k (bound kernel 2.31 -> 1.92 ms per
#ifndef PR_RING_OFF                                             // group-step, the walk as before); 6 / 2 41.9 k, 7 / 2 40.0 k; without the leading row -- 4 / 0, 5 / 0 -- or the last child -- 4 / 1 -- the
#define PR_RING_OFF 1                                           // bound gets loose and the walk pays: 34.1 / 33.4 / 34.7 k
#endif
#ifndef PR_WIN_ROWS
#define PR_WIN_ROWS 1                                           // rows of a pixel window loaded before the first is looked at (beyond the 3 x 3 case, which is one round trip anyway)
#endif
#ifndef PR_NN_WINFIRST
#define PR_NN_WINFIRST 4.0e-4f                                  // (20 mm)^2 [same box: off 47.4 k, (2 mm)^2 47.4 k, (3 mm)^2 47.8 k, (5 mm)^2 48.2 k, (10 mm)^2 48.8 k, (20 mm)^2 49.3 k, always 48.9 k poses/s]: a query whose previous winner is within this (but which moved too far to go without a new bound) scans the largest window BEFORE any descent
#endif
#ifndef PR_NN_WINFIRST_EARLY
#define PR_NN_WINFIRST_EARLY 4.0e-6f                            // the same threshold in pass 1, where the cloud has just made its largest step and 87 % of the attempts would fail: (2 mm)^2
#endif
#ifndef PR_NN_COVER_PAD
#define PR_NN_COVER_PAD 5.0e-4f                                 // metres a window search looks beyond its bound for the runner-up (about one pixel at 300 mm)
#endif
#ifndef PR_NN_SETTLE
#define PR_NN_SETTLE 1                                          // 1: a point that has stopped moving takes the widest cover the window holds (its margin then lasts for the rest of the loop)
#endif
#ifndef PR_NN_STILL
#define PR_NN_STILL 2.5e-7f                                     // (0.5 mm)^2: below this step a point's previous winner is taken as a tight seed
#endif
#ifndef PR_NN_NODESCENT
#define PR_NN_NODESCENT 2.5e-7f                                 // squared step up to which the previous winner's distance is bound enough (no descent through the representatives)
#endif

// ---- kd-tree search: task walk over wide nodes (nn_search.hip) ----------------------------------------------------------------------
// Round 4 sweep, same box, 256 hypotheses, sum of the walk over the 21 passes: workgroups per hypothesis 5 / 8 / 12 / 16 / 32 -> 5.98 / 5.06 /
// 4.58 / 4.91 / 5.46 ms (the bound kernel follows: 2.31 / 2.16 / 1.92 / 1.97 / 2.04); at 12, queue capacities and waves per SIMD (384, 288, 5) /
// (256, 192, 6) / (320, 224, 6) / (224, 160, 7) / (192, 160, 6) / (160, 128, 8) -> 4.60 / 4.27 / 4.35 / 4.86 / 6.55 / 8.01 ms: more waves in flight
// help every pass, but the first pass needs its queues (overflows are walked a second time).  PIPELINED (configs[2] through the two slots, two
// pose groups, same box): 8 / 12 / 16 / 18 / 20 / 24 / 28 / 36 workgroups -> 32.6 / 34.6-35.0 / 33.5 / 35.6 / 35.6 / 35.3 / 36.0 / 35.8 k poses/s.
// After the leaf-step change (same box, four runs each): capacities 256 / 192 (rounds 3-4) 43.2 k, 256 / 160 43.3 k, 240 / 176 41.9 k, 224 / 160 40.7 k poses/s;
// two more boxes, eight runs each: 256 / 192 43.0 k, **288 / 160 43.8 k** (the same LDS), 304 / 144 43.9 k, 320 / 128 43.5 k, 352 / 96 43.1 k, 320 / 160 43.6 k.
#ifndef PR_TREE_GX
#define PR_TREE_GX 20                                           // workgroups per hypothesis of the bound kernel and the walk (more for launches with few hypotheses)
#endif
#ifndef PR_WIDE_WAVES
#define PR_WIDE_WAVES 6                                         // wavefronts per SIMD the task walk is compiled for (<= 80 VGPRs; LDS: 26.5 KiB per workgroup)
#endif
#ifndef PR_WIDE_LANES
#define PR_WIDE_LANES 2                                         // lanes per task: the paired record layout is made for 2
#endif
#ifndef PR_WIDE_LEAF_PER
#define PR_WIDE_LEAF_PER 5                                      // points of a leaf a lane tests per round: 2 x 5 = the reference's max_leaf in ONE round (the records allow 15-point leaves: two)
#endif
#ifndef PR_KD_PER
#define PR_KD_PER 4                                             // kd-tree build: consecutive positions per thread of the per-level passes (tile = 256 x this)
#endif
#ifndef PR_WIDE_QCAP
#define PR_WIDE_QCAP 288                                        // entries of a wavefront's node-task queue
#endif
#ifndef PR_WIDE_LCAP
#define PR_WIDE_LCAP 160                                        // entries of a wavefront's leaf-task queue
#endif
