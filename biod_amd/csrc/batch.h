// batch.h -- the state of a batched pass (batch.cpp), shared with the file feeder (stream.cpp)
#pragma once
#include "common.h"

struct Batch {
    uint64_t batch_bytes = 0, reserve = 0;
    int want_entries = 0; uint32_t window = 65536;
    // Three pinned buffers: one takes the caller's bytes, full ones wait in a queue for a free device buffer, the rest are free or the
    // source of a copy in flight (host_free tells).  With two, the full buffer a caller commits while batch i + 1 is on its way could
    // only leave at the caller's next call -- after it had been handed batch i + 1's windows -- and the two directions of the link took turns.
    static constexpr int NH = 3;
    HostBuf host[NH]; uint64_t host_len[NH] = {0, 0, 0}; hipEvent_t host_free[NH] = {nullptr, nullptr, nullptr};
    int fill = 0;                                       // host buffer taking the pushes (-1: none to be had before a full one has left)
    int fullq[NH] = {0, 0, 0}; int n_full = 0;          // full buffers waiting for a device buffer, oldest first
    int acq = -1; uint64_t acq_off = 0;                 // bgp_batch_acquire handed out host[acq] from acq_off on and waits for its commit (-1: nothing outstanding)
    // Device buffers: two (one in its kernels, one loading).  A file-fed pass (stream.cpp) can be given more (BGP_STREAM_BUFFERS, up to
    // NDV) -- measured and not the default: k_inflate's waves live ten milliseconds and take every wave slot and all of a CU's LDS they can
    // get, so a batch's kernels launched beside two or three inflating batches wait for slots (12 ms a 256 MiB batch instead of 1.5).
    // What works is ONE large inflate at a time, launched when the batch's kernels are through, beside nothing but the windows' copies.
    static constexpr int NDV = 4; int n_dev = 2;
    DevBuf dev[NDV]; uint64_t new_off[NDV] = {0, 0, 0, 0}, new_len[NDV] = {0, 0, 0, 0}; hipEvent_t loaded_ev[NDV] = {nullptr, nullptr, nullptr, nullptr};
    enum { FREE, LOADED, CURRENT } state[NDV] = {FREE, FREE, FREE, FREE};
    int queue[NDV] = {0, 0, 0, 0}; int n_queue = 0;     // loaded device buffers, oldest first
    hipStream_t copy_stream = nullptr;
    bool ended = false, any_batch = false;
    int cur = -1; bool cur_final = false;
    uint64_t carry_off = 0, carry_len = 0;              // suffix of the current batch's stream the next batch begins with
    uint64_t ord_base = 0, next_ord_base = 0;           // stream-wide ordinal of the current batch's record 0
    bool resume = false; int32_t resume_ref = 0; uint64_t resume_pos = 0;      // where handing out stopped before this batch
    bool cut = false; int32_t cut_ref = 0; uint64_t cut_pos = 0;                // where it stops in this one
    uint32_t run = 0; uint64_t next_pos = 0; bool run_open = false;
    // use_md_tag across batches: the provider chain's state at the last event in front of the cut (stream-wide record ordinals)
    struct { bool have = false; uint64_t P_ord = 0, i_ord = 0; uint32_t ci = 0, y = 0, ce = 0; int32_t ref = -1; } md;
    const uint8_t* pin_ptr = nullptr; uint64_t pin_len = 0;      // bgp_batch_push_pinned: page-locked bytes of the caller's still to go to the device (a device buffer at a time, as buffers fall free)
    struct FileFeed* feed = nullptr;                    // bgp_stream_bam_begin (stream.cpp): the batches come from a BGZF file through k_inflate, nobody pushes
};

// stream.cpp: the feeder of bgp_stream_bam_begin.  feed_pump enqueues the next batches (upload events -> k_inflate into a free device
// buffer -> LOADED) while device buffers are free; `need_one`: wait for the file until one batch is queued or the file is over (then
// b->ended).  feed_check: the inflate's per-block status of the batch in dev[q], once its event has passed.  feed_release: stop the readers.
int batch_begin_core(bgp_ctx* c, uint64_t batch_bytes, int want_entries, uint32_t window_positions, bool host_buffers);      // batch.cpp
int feed_pump(bgp_ctx* c, Batch* b, bool need_one);
int feed_check(bgp_ctx* c, Batch* b, int q);
void feed_release(bgp_ctx* c, Batch* b);
