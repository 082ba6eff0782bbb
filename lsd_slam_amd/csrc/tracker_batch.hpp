// Interface between tracker.hip (host side of the SE3 tracker) and tracker_batch.hip (throughput-mode evaluation kernel).
#pragma once
#include "track_device.hpp"
// evaluates the state st2[job][parity] of every job of the batch (grid strips x n jobs) into scratch[job][parity]
int lsd_track_eval_tiles_launch(lsdhip_tracker* t, int grid, int n, const TrackScratch& sc, int parity);
