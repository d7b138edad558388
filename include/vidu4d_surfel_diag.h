/*
 * vidu4d_surfel_diag.h -- diagnostics of libvidu4d_surfel.so: NOT part of the drop-in boundary (vidu4d_surfel.h).
 * Nothing here is needed to run the path; bench.py uses the stage timers for its roofline leg and the walk counters
 * for the lane-utilisation note, tools/ use both.  The reference has no equivalent (its only timing aid is
 * torch.profiler around whole steps, lab4d/utils/profile_utils.py:113-161).
 */
#ifndef VIDU4D_SURFEL_DIAG_H_INCLUDED
#define VIDU4D_SURFEL_DIAG_H_INCLUDED

#ifdef __cplusplus
extern "C" {
#endif

/* ---- per-stage timing with HIP events recorded on the launch stream (off by default).  Used by
 *      bench.py to measure each kernel's average launch duration inside the timed region; the
 *      reference has no equivalent (its only timing aid is torch.profiler around whole steps,
 *      lab4d/utils/profile_utils.py:113-161). ---- */
int vidu4d_surfel_profile_enable(int on);
/* Tile-walk counters: a vidu4d_surfel_backward call whose args->diag_walk_counters points at VIDU4D_BLEND_STATS device
 * u64 counters (zeroed by the caller; per call, no library state) also counts what its tile walk looks like, by a
 * counting kernel queued behind the backward blend (outside the stage timers' blend_bwd span): [0] list entries staged, [1] wave
 * trips (one pass of a wave through the walk's loop body: one list entry per 32-lane half), [2] those with a contributing
 * lane, [3] contributing lanes = (pixel, entry) pairs, [4] 16-lane rows with a contributing lane, [5..9] trips of [2] with
 * <= 4 / 8 / 16 / 32 / 64 contributing lanes, [10] trips of [1] in which no lane passes the pair test (the cull test's
 * footprints reach the two 8x4 pixel blocks, the exact ones do not).  [3] / (64 [1]) is the
 * lane utilisation bench.py's roofline note quotes. */
#define VIDU4D_BLEND_STATS 11
int vidu4d_surfel_profile_stage_count(void);
const char* vidu4d_surfel_profile_stage_name(int stage);
int vidu4d_surfel_profile_read(double* total_ms /*[stage_count]*/, long long* count /*[stage_count]*/, int reset);

/* A device-to-device copy of `bytes` (16-byte aligned pointers), `repeat` times over, by `workgroups` workgroups of 512
 * threads: the footprint of a collective's kernel, for measuring what such a co-tenant costs the blend kernels
 * (tools/contention_probe.py -> profiles/r04_contention.json, read by bench.py's scaling model). */
int vidu4d_diag_copy(void* dst, const void* src, size_t bytes, int workgroups, int repeat, void* stream);

#ifdef __cplusplus
}
#endif
#endif
