/* lsdhip_driver.h — C entry points of liblsdhip_driver.so: the C++ sequence driver (lsd_slam_hip::SlamLoop in
 * include/lsd_slam_hip.hpp, i.e. the doSlam=false / blockUntilMapped=true slice of SlamSystem, C/SlamSystem.cpp:890-1040,
 * :739-828, :542-614, :458-490) behind plain C so that bench.py and tests can run whole frame batches without going
 * through the Python interpreter per frame.  Not part of the drop-in boundary (that is include/lsdhip.h). */
#ifndef LSDHIP_DRIVER_H
#define LSDHIP_DRIVER_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif
typedef struct lsdloop lsdloop;
typedef struct lsdloop_stats {
  double seconds_track;      /* wall time of the batches minus the two below: frame creation + SE3Tracker::trackFrame */
  double seconds_map;        /* GPU time (HIP events) of DepthMap::updateKeyframe (frames that did not become keyframes) */
  double seconds_keyframe;   /* GPU time of finalizeKeyFrame + createKeyFrame (frames that became keyframes) */
  long long frames;          /* frames tracked */
  long long updates;         /* updateKeyframe calls */
  long long keyframes;       /* keyframes created */
  long long evaluations;     /* residual evaluations (k_track_step launches that did work) */
  long long tracked_good;    /* frames whose trackFrame ended with trackingWasGood (C/Tracking/SE3Tracker.cpp:472-477) */
  long long level_evaluations[5];   /* residual evaluations per pyramid level 0..4 */
  long long track_launches;  /* k_track_step launches of the tracking jobs that evaluated */
  long long dropped;         /* pipelined loops: frames tracked on a keyframe the mapper had already replaced, hence not mapped (C/SlamSystem.cpp:559-566) */
} lsdloop_stats;
/* K4 = fx, fy, cx, cy.  first_image / images: uint8 w*h, host memory or (images_on_device != 0) memory of `device`. */
int lsdloop_create(int device, int w, int h, const float K4[4], const uint8_t* first_image, int images_on_device,
                   const float* gt_depth0_host, int kf_every, lsdloop** out);
void lsdloop_destroy(lsdloop* l);
/* Runs up to n frames; stops early right after a frame that became a keyframe when stop_at_keyframe != 0.
 * Returns the number of frames consumed (>= 0) or a negative lsdhip status.  frameToKeyframe_out: n x 7 doubles or NULL. */
int lsdloop_run(lsdloop* l, const uint8_t* const* images, int n, int stop_at_keyframe, double* frameToKeyframe_out);
int lsdloop_get_stats(lsdloop* l, lsdloop_stats* out);
int lsdloop_reset_stats(lsdloop* l);
/* smoothed idepth / variance planes of the current keyframe, device to device (the multi-GPU gather payload) */
int lsdloop_copy_keyframe_planes(lsdloop* l, float* idepth_dev, float* idepthVar_dev);
/* Export of finished keyframes without leaving the loop: from now on every keyframe that is finalised has its (idepth,
 * idepthVar) level-0 planes copied (device to device, on the loop's stream) into slot (count % slots) of ring_dev, a device
 * buffer of slots x 2 x w x h floats; count restarts at 0.  ring_dev = NULL switches the export off.
 * lsdloop_keyframes_exported returns count.  The copies are complete when lsdloop_run returns. */
int lsdloop_set_keyframe_ring(lsdloop* l, float* ring_dev, int slots);
long long lsdloop_keyframes_exported(lsdloop* l);
/* Validation helpers (bench.py's self-check): keep every keyframe created from now on (on != 0) so that, after the run, the
 * Sim3 scale createKeyFrame gave each of them (C/DepthEstimation/DepthMap.cpp:1286-1305) and its semi-dense point count can
 * be read back: scales_out / points_out receive up to max entries in creation order; returns the number of keyframes kept.
 * The read-back resolves deferred results, i.e. synchronises — call it outside timed regions. */
int lsdloop_keep_keyframes(lsdloop* l, int on);
int lsdloop_keyframe_log(lsdloop* l, double* scales_out, long long* points_out, int max);
/* on != 0: tracking beside mapping (the reference's two threads with blockUntilMapped == false, C/SlamSystem.cpp:1026-1040), the mapper
 * exactly one frame behind the tracker: see lsd_slam_hip::SlamLoop.  Call before the first lsdloop_run.  Default 0 (blockUntilMapped). */
int lsdloop_set_pipeline(lsdloop* l, int on);
/* frames handed to DepthMap::updateKeyframe per mapping iteration (1 = blockUntilMapped, the default; K > 1 restates live
 * operation where the mapper finds up to K tracked frames queued, C/SlamSystem.cpp:559-571) */
int lsdloop_set_live_queue(lsdloop* l, int frames);
/* LM retries evaluated per k_track_step launch (lsdhip_tracker_set_speculation); default 5, 1 = one evaluation per launch. */
int lsdloop_set_speculation(lsdloop* l, int trials, int finest_level_workgroups);
/* lsdhip_depth_observe_time of the loop's depth map */
int lsdloop_observe_time(lsdloop* l, double* ms_out, long long* calls_out);
/* lsdhip_depth_observe_work of the loop's depth map: launches counted, searched pixels, walk steps */
int lsdloop_observe_work(lsdloop* l, double out3[3]);
/* ---- RCCL over xGMI, issued from the C++ loop on its own stream (BASELINE.json configs[3]: independent sequences, one per GPU,
 * results collected on rank 0).  The library binds the ncclXxx entry points of the RCCL already loaded into the process (the one
 * torch.distributed uses) or of librccl.so, by name.  Rendezvous: rank 0 calls lsdloop_comm_unique_id and hands the 128 bytes to
 * the other ranks by any means (bench.py: torch.distributed object broadcast); every rank then calls lsdloop_comm_init.
 * lsdloop_gather_keyframes enqueues, on the loop's stream and without any host synchronisation, the collection of the first
 * `count` ring slots (lsdloop_set_keyframe_ring) of every rank into recv_dev on rank `root`: recv_dev holds world x stride_floats
 * floats, rank r's slots land at recv_dev + r * stride_floats (ncclGroupStart; ncclSend / ncclRecv per peer; ncclGroupEnd; the
 * root's own slots are a device copy).  All ranks pass the same count.  Returns 0, or a negative status (lsdloop_last_error). */
int lsdloop_comm_unique_id(unsigned char out128[128]);
int lsdloop_comm_init(lsdloop* l, const unsigned char id128[128], int rank, int world);
int lsdloop_comm_destroy(lsdloop* l);
int lsdloop_gather_keyframes(lsdloop* l, int count, int root, float* recv_dev, long long stride_floats);
/* every rank's keyframe count of the last RCCL gather — the counts travel ahead of the planes (one int per rank, ncclAllGather) and
 * size the root's receives, so ranks whose sequences finished different numbers of keyframes cannot desynchronise the gather */
int lsdloop_gather_counts(lsdloop* l, int* counts_out, int cap);
/* Second transport of the gather for the processes of one node (no RCCL; e.g. two processes on one GPU): the root owns an IPC mailbox
 * of world x ring_slots keyframe slots plus flags; rank r copies its new ring slots into slot block r, publishes its count and raises
 * its ready flag; lsdloop_gather_keyframes then takes this path (root / recv_dev / stride arguments are ignored).
 *   lsdloop_ipc_init (after lsdloop_set_keyframe_ring): the root writes its 64-byte handle, the others zeros;
 *   lsdloop_ipc_connect: the root's handle;
 *   lsdloop_ipc_result: synchronises; on the root counts_out[world] and the device pointer of the [world][ring_slots][2][h][w] floats;
 *     returns 0, or the flag value a bounded wait gave up on. */
int lsdloop_ipc_init(lsdloop* l, int rank, int world, int root, unsigned char handle64_out[64]);
int lsdloop_ipc_connect(lsdloop* l, const unsigned char root_handle64[64]);
int lsdloop_ipc_result(lsdloop* l, int* counts_out, float** data_out);
void* lsdloop_ctx(lsdloop* l);   /* the lsdhip_ctx* the loop runs on (prof hooks, stream) */
/* ---- S sequences sharing one GPU (lsd_slam_hip::SlamLoopBatch; BASELINE.json configs[3] with more sequences than GPUs): per sequence
 * the blockUntilMapped loop above, across sequences every stage in shared launches (lsdhip_frame_create_batch,
 * lsdhip_tracker_track_batch, lsdhip_depth_update_batch).  first_images / gt_depth0_host: S pointers.
 * lsdloopbatch_run: n steps = n frames of every sequence, images[t * S + s]; frameToKeyframe_out: n x S x 7 doubles or NULL; returns
 * the number of steps run.  lsdloopbatch_get_stats: 6 values per sequence (frames, tracked_good, updates, keyframes, evaluations,
 * lost 0/1). */
typedef struct lsdloopbatch lsdloopbatch;
int lsdloopbatch_create(int device, int w, int h, const float K4[4], int S, const uint8_t* const* first_images, int images_on_device,
                        const float* const* gt_depth0_host, int kf_every, lsdloopbatch** out);
void lsdloopbatch_destroy(lsdloopbatch* l);
int lsdloopbatch_run(lsdloopbatch* l, const uint8_t* const* images, int n, double* frameToKeyframe_out);
int lsdloopbatch_get_stats(lsdloopbatch* l, long long* out6_per_sequence);
void* lsdloopbatch_ctx(lsdloopbatch* l);
/* phase[s] in [0, kf_every): sequence s changes keyframe as if its current keyframe were already phase[s] frames old (sequences that
 * start together otherwise all change keyframe in the same step, which independent cameras do not).  Before the first run. */
int lsdloopbatch_set_keyframe_phases(lsdloopbatch* l, const int* phase);
/* on != 0: tracking beside mapping for all sequences (lsd_slam_hip::SlamLoopBatch::setPipelined; per sequence the schedule of
 * lsdloop_set_pipeline: mapper one frame behind, frames tracked on a replaced keyframe dropped).  Before the first run.
 * lsdloopbatch_dropped counts the dropped frames of a sequence. */
int lsdloopbatch_set_pipeline(lsdloopbatch* l, int on);
/* lsdhip_tracker_set_batch_coarse_min_jobs of the loop's tracker (sequences per step from which the coarse pyramid levels of a tracking
 * batch run in one workgroup per sequence; 0: never) */
int lsdloopbatch_set_coarse_min_jobs(lsdloopbatch* l, int min_jobs);
long long lsdloopbatch_dropped(lsdloopbatch* l, int sequence);
/* Read-outs per sequence, for validation against the reference's single-sequence loop (tests/test_multiseq_gpu.py): keep every promoted
 * keyframe alive and report its createKeyFrame rescale factor / point count (as lsdloop_keep_keyframes / lsdloop_keyframe_log); the
 * last trackFrame result of a sequence; its current depth map in the reference's 32-byte hypothesis layout (w x h entries). */
int lsdloopbatch_keep_keyframes(lsdloopbatch* l, int on);
int lsdloopbatch_keyframe_log(lsdloopbatch* l, int sequence, double* scales_out, long long* points_out, int max);
int lsdloopbatch_last_result(lsdloopbatch* l, int sequence, lsdhip_track_result* out);
int lsdloopbatch_download_map(lsdloopbatch* l, int sequence, lsdhip_hypothesis* out);   /* frames of a sequence tracked on a replaced keyframe (pipelined): not mapped */
const char* lsdloop_last_error(void);
/* ---- row-band decomposition of the regulariser (BASELINE.json configs[4]) ----------------------------
 * `world` bands over an H-row map; this process holds bands [first_band, first_band + n_local) as windows of
 * lsdband_window_rows() rows on `device` (index arithmetic: lsd_slam_amd/bands.py BandPlan).  lsdband_run queues `passes`
 * fused fill-holes + regularise passes over every local window with a halo refresh between passes and returns without
 * waiting: windows of this process refresh each other with one map -> map copy launch, rows of other processes travel packed
 * through ncclGroupStart / ncclSend / ncclRecv / ncclGroupEnd on the context's stream (lsdband_comm_init: communicator over
 * processes from an ncclUniqueId, lsdloop_comm_unique_id makes one).  hyp rows use lsdhip_hypothesis (32 bytes per pixel). */
typedef struct lsdband lsdband;
int lsdband_create(int device, int w, int H, int world, int first_band, int n_local, lsdband** out);
/* the index arithmetic alone (no GPU): layout4[4 r ..] = owned [y0, y1), window [a, b) of band r; segments4 = (receiving band,
 * owning band, first global row, rows) per halo segment, at most cap of them written; returns the number of segments */
int lsdband_plan(int H, int world, int* window_rows_out, int* layout4, int* segments4, int cap);
void lsdband_destroy(lsdband* b);
int lsdband_window_rows(const lsdband* b);
int lsdband_layout(const lsdband* b, int band, int out4[4]);   /* owned [y0, y1), window [a, b) */
int lsdband_load(lsdband* b, int local, const void* hyp_window, const float* maxgrad_window);
int lsdband_get(lsdband* b, int local, void* hyp_window_out);   /* synchronises */
int lsdband_comm_init(lsdband* b, const void* unique_id128, int nprocs, int proc, const int* proc_of_band);
int lsdband_run(lsdband* b, int passes);
int lsdband_synchronize(lsdband* b);
long long lsdband_halo_bytes_per_pass(const lsdband* b);
/* test hook: windows of this process also exchange through the packed path (pack launch, device copy in place of
 * ncclSend / ncclRecv, unpack launch) — RCCL refuses two ranks on one GPU, so this is how a one-GPU box exercises it */
/* Second transport between the processes of one node, without RCCL (which refuses two ranks on one device): every process exports
 * one IPC "mailbox" allocation (flags + two packed receive buffers per incoming remote segment) and maps the others'; a sender packs
 * its halo rows straight into the receiver's buffer and raises the segment's `ready` flag, the receiver unpacks and raises `consumed`.
 * Same pack -> transfer -> unpack schedule per pass as the RCCL path, all of it queued on the context's stream.
 *   lsdband_ipc_init: band ownership as lsdband_comm_init; writes this process's 64-byte handle;
 *   lsdband_ipc_connect: the handles of all processes (nprocs x 64 bytes, process order);
 *   lsdband_ipc_failed (after lsdband_synchronize): 0, or the flag value a bounded wait gave up on. */
int lsdband_ipc_init(lsdband* b, int nprocs, int proc, const int* proc_of_band, unsigned char handle64_out[64]);
int lsdband_ipc_connect(lsdband* b, const unsigned char* handles);
int lsdband_ipc_failed(lsdband* b);
/* index arithmetic of the overlapped pass, no GPU needed: the tile rows (8 map rows each, window coordinates) of band `band` that hold
 * owned rows, as runs (first tile row, tile rows, edge) -> 3 ints each; edge = 1: issued before the exchange forks */
int lsdband_tile_runs(int H, int world, int band, int* runs3, int cap);
int lsdband_set_packed_exchange(lsdband* b, int on);
/* 1 (default): when the object exchanges rows with other processes, a pass is issued as edge tile rows + interior tile rows and the
 * exchange runs on the context's transport stream under the interior part; 0: one launch per window and pass, then the exchange */
int lsdband_set_overlap(lsdband* b, int on);

#ifdef __cplusplus
}
#endif
#endif
