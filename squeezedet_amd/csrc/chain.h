// Riders of the fire_chain launches (chain.hip): the serving step's decode + filter of the PREVIOUS batch.
//
// Every launch of the SqueezeDet forward fills the chip exactly once -- persistent kernels with a static share of tiles per
// workgroup, or one workgroup per CU -- so side work on another stream costs a whole "round" of whatever it lands beside
// (measured: 35 us per 0.49 ms step wherever the filter launch was placed), and every event that orders the two streams
// drains the forward's queue for ~6 us.  The six fire_chain launches are the exception: 240 workgroups of one per CU at
// batch 32, i.e. 2 idle CUs per XCD.  A launch given a ChainRide appends `nriders` workgroups (spread over the XCDs, behind
// the chain workgroups in dispatch order) that run filter_body.h's per-image body for images [img0, img0 + nimg) of the
// previous batch and write the rows where the caller wants them (pinned host memory).  Same stream as the forward: stream
// order alone makes the previous ConvDet's preds / scores visible and keeps the slot's buffers from being overwritten early.
#pragma once
#include "postproc.h"

namespace sqdet {

struct ChainRide {
  FilterArgs fa;       // a.probs = the previous batch's scores [n, A]; out_* rows of ALL its images (indexed by image)
  DecodeArgs da;       // its preds, the anchors
  int img0, nimg;      // images this launch takes
  int nriders;         // workgroups that share them (<= fire_chain_idle_cus)
};

// CUs a fire_chain launch of this shape leaves idle (0: none, or the shape runs as the persistent form)
int fire_chain_idle_cus(int n, int h, int w, int s1x1, int e1x1, int e3x3, int next_s1x1, int dtype);
// sqdet_fire_chain_fwd with riders (ride == NULL: none)
int fire_chain_launch_ride(const void* sq_in, const void* stream_buf, const float* b_e1, const float* b_e3,
                           const float* b_next_s, void* y, void* sq_out, int n, int h, int w, int s1x1, int e1x1, int e3x3,
                           int next_s1x1, int dtype, const ChainRide* ride, hipStream_t st);

}  // namespace sqdet
