/*
 * Bit-rate control of one MJPEG stream: which JPEG quality the next frame is
 * coded with, whether it is coded at all, and how the stream's byte budget
 * reacts to client reports and server-side drops.  Host-only policy; it sees
 * frame sizes and clocks, never pixels.  Behaviour follows
 * server/mjpeg_encoder.c:362-727 and :886-1258 decision for decision, so a
 * stream driven by the same sizes, clocks and reports takes the same path.
 */
#ifndef SPICE_AMD_MJPEG_RATE_H
#define SPICE_AMD_MJPEG_RATE_H

#include <stdint.h>

#include "../../include/spice_gpu.h"

namespace spicegpu {

/* JPEG qualities a stream moves between (server/mjpeg_encoder.c:30-33) */
enum { MJPEG_RATE_LADDER = 7, MJPEG_RATE_STATIC_RUNG = 5 };
extern const int mjpeg_rate_quality[MJPEG_RATE_LADDER];

class MjpegRate {
public:
    typedef uint64_t (*ClockFn)(void *opaque);

    /* active == false: fixed quality 70 at 25 fps, no gating (:213) */
    void init(bool active, uint64_t starting_bit_rate, const MJpegEncoderRateControlCbs *cbs,
              void *cbs_opaque);
    void set_clock(ClockFn fn, void *opaque);

    bool active() const { return active_; }
    int quality() const { return mjpeg_rate_quality[rung_]; }
    uint64_t bit_rate() const { return byte_rate_ * 8; }

    /* Before a frame is coded (:736-769).  False = drop this frame. */
    bool admit_frame(uint32_t frame_mm_time);
    /* After a frame was coded into `size` bytes (:865-882) */
    void frame_coded(uint64_t size);
    /* A scanline could not be written (:848-851) */
    void frame_aborted() { last_size_ = 0; }

    void client_report(uint32_t num_frames, uint32_t num_drops, uint32_t start_mm_time,
                       uint32_t end_mm_time, int32_t end_frame_delay, uint32_t audio_delay);
    void server_dropped_frame();

    void snapshot(SpiceGpuMjpegRateState *out) const;

private:
    enum Ladder { SET = 0, UPGRADE = 1, DOWNGRADE = 2 };
    enum Cause { SIZE_CHANGE = 0, RATE_CHANGE = 1 };

    struct Probe {                       /* one round of ladder probing */
        int kind, cause;
        uint64_t size_at[MJPEG_RATE_LADDER];
        int floor_rung, floor_fps;
        int ceil_rung, ceil_fps;
        int best_fps, best_fps_rung;
    };
    struct Budget {                      /* what was sent since the byte rate last changed */
        uint64_t since_ns, last_frame_ns;
        uint32_t since_mm;
        int raised;
        uint32_t frames;
        uint64_t bytes;
    };

    uint64_t now() const;
    uint32_t latency_ms() const;
    uint32_t source_fps() const;
    void settle(int rung, uint32_t fps, uint64_t frame_size);
    void probe_step();
    void finish_probe();
    void begin_upgrade(int cause, int floor_rung, int floor_fps);
    void begin_downgrade(int cause, int ceil_rung, int ceil_fps);
    void cancel_probe();
    void track_frame_size();
    void retune_fps(uint64_t t);
    void lower_budget();
    void raise_budget();
    void negative_report(uint32_t end_mm_time);
    void positive_report(uint32_t start_mm_time);
    void weigh_server_drops();
    void forget_budget_history(int raised);

    bool active_ = false;
    MJpegEncoderRateControlCbs cbs_ = {nullptr, nullptr, nullptr};
    void *cbs_opaque_ = nullptr;
    ClockFn clock_ = nullptr;
    void *clock_opaque_ = nullptr;

    bool probing_ = false;
    Probe probe_ = {};
    Budget budget_ = {};
    int peak_video_delay_ = 0;
    uint32_t peak_audio_delay_ = 0;
    uint32_t coded_ = 0, dropped_ = 0;   /* server side, since last weighed */

    uint64_t byte_rate_ = 0;
    int rung_ = 0;
    uint32_t fps_ = 0;
    double paced_fps_ = 0;               /* the rate frames are admitted at */
    uint64_t paced_since_ns_ = 0;
    uint64_t paced_frames_ = 0;
    uint64_t settled_size_ = 0;
    uint64_t last_size_ = 0;
    uint64_t recent_bytes_ = 0;
    uint32_t recent_frames_ = 0;
    uint64_t warmup_from_ns_ = 0;

    mutable uint32_t delay_updates_ = 0, delay_last_ = 0;
    void tell_playback_delay(uint32_t ms) const;
};

} // namespace spicegpu

#endif
