// Interface between tracker.hip (host side of the SE3 tracker) and tracker_coarse.hip (the cluster kernel).
#pragma once
#include "lsdhip_internal.hpp"
#define CT_BLOCK 512
#define CT_POOL 6656            // most pixels of the cluster levels of a job together (and of any one of them)
#define CT_GMAX 13              // strips (workgroups) per trial: CT_POOL / CT_BLOCK
#define CT_LEVELS 3             // most levels of a job inside the kernel
#define CT_ROW 168              // granules per row: [0, 41) partial sums | [41, 44) keys | [44, 47) K2 error of the keys' points |
#define CT_KEY0 41              //   [48, 73) record: inc[6], Tn (q, t)[7], R[9], t[3] (strip 0 only) | [80, 167) 3 x 29 contributions
#define CT_WSUB0 44
#define CT_REC0 48
#define CT_SUB0 80
#define CT_HEAD 47              // granules of a row every workgroup reads
#define CT_SPIN_LIMIT (1u << 19)     // polls of one lane before the kernel gives up (~1 s)
typedef unsigned long long ct_u64;
typedef __attribute__((address_space(1))) ct_u64 ct_gu64;
struct CoarsePlan {
  int nt;                    // trials per set at most (the launch has nt x gmax workgroups)
  int gmax;                  // strips of the largest cluster level
  int low;                   // lowest level that runs in this kernel (> job.lastLevel)
  int trials[LSD_LEVELS];    // trials per set at each level (<= nt)
};

inline size_t lsd_track_coarse_rows_bytes() { return (size_t)2 * LSD_SPEC_MAX * CT_GMAX * CT_ROW * sizeof(ct_u64); }
int lsd_track_coarse_launch(lsdhip_tracker* t, const TrackJob& job, const CoarsePlan& plan);
