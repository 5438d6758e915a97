// pr_tuning.h -- the compile-time knobs of the kernels, in one place.  Each is a NUMBER that was swept on hardware (DESIGN.md / profiles/NOTES.md hold the
// sweeps); -DPR_X=... on the hipcc line (PR_EXTRA_FLAGS of pose_refine_amd/build.py) overrides one for an A/B build (tools/ab_flags.sh).
// The on/off switches of rounds 1-3 whose A/B is decided are gone: their winning branch is the code.
#pragma once

#ifndef PR_LEAF_BATCH
#define PR_LEAF_BATCH 10
#endif
#ifndef PR_NN_WIDE_BOUND
#define PR_NN_WIDE_BOUND 4.0e-6f                                // (2 mm)^2: above it a node's whole record is fetched at once
#endif
#ifndef PR_NN_COVER_PAD
#define PR_NN_COVER_PAD 5.0e-4f                                 // metres a window search looks beyond its bound for the runner-up (about one pixel at 300 mm)
#endif
#ifndef PR_NN_NODESCENT
#define PR_NN_NODESCENT 2.5e-7f                                // squared step up to which the previous winner's distance is bound enough (no descent through the representatives)
#endif
#ifndef PR_RING_ROWS
#define PR_RING_ROWS 2
#endif
#ifndef PR_NN_SETTLE
#define PR_NN_SETTLE 1                                           // grid_search: widest cover for points that have stopped moving
#endif
#ifndef PR_NN_STILL
#define PR_NN_STILL 2.5e-7f                                     // (0.5 mm)^2: below this step a point's previous winner is taken as a tight seed
#endif
#ifndef PR_GRID_MAXW
#define PR_GRID_MAXW 2
#endif
#ifndef PR_PASS_WAVES
#define PR_PASS_WAVES 1
#endif
#ifndef PR_GATHER_BATCH
#define PR_GATHER_BATCH 4
#endif
#ifndef PR_WIDE_WAVES
#define PR_WIDE_WAVES 5                                        // wavefronts per SIMD the task walk is compiled for
#endif
#ifndef PR_WIDE_LANES
#define PR_WIDE_LANES 2                                        // lanes per task
#endif
#ifndef PR_WIDE_QCAP
#define PR_WIDE_QCAP 384
#endif
#ifndef PR_WIDE_LCAP
#define PR_WIDE_LCAP 288
#endif
#ifndef PR_TREE_GX
#define PR_TREE_GX 8
#endif

#define PR_SEL_DOC 0   /* (PR_SEL / PR_TREE_LEVEL / PR_HD are local helper macros of their files, not knobs) */
