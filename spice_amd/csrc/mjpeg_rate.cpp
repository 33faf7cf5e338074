/*
 * MJPEG stream bit-rate control (host policy; see mjpeg_rate.h).
 *
 * Vocabulary: the seven JPEG qualities form a ladder of rungs 0..6; "probing"
 * is the reference's quality evaluation (successive frames coded on different
 * rungs until the best rung the byte budget can carry is known); "settling"
 * ends a probe and fixes rung and fps.
 *
 * Integer/double conversions are written out where the reference relies on C's
 * implicit ones, because the decisions depend on them.
 */
#include "mjpeg_rate.h"

#include <string.h>
#include <time.h>

namespace spicegpu {

const int mjpeg_rate_quality[MJPEG_RATE_LADDER] = {20, 30, 40, 50, 60, 70, 80};

namespace {

/* server/mjpeg_encoder.c:27-69 */
const uint32_t FPS_CEIL = 25, FPS_FLOOR = 1;
const uint32_t FPS_STRICT = 10, FPS_LENIENT = 5;
const uint32_t SIZE_WINDOW = 3;
const uint32_t BUDGET_MIN_FRAMES = 3;
const double SERVER_DROP_SHARE = 0.1;
const int REPORT_WAIT_MM = 2000, REPORT_WAIT_STRICT_MM = 3000;
const uint64_t PACE_REVIEW_MS = 500;
const uint32_t PLAYBACK_DELAY_CAP_MS = 5000;
const uint64_t WARMUP_NS = 3000ull * 1000 * 1000;
const int MID_RUNG = MJPEG_RATE_LADDER / 2;
/* server/mjpeg_encoder.c:1110-1112 */
const double LATENCY_HALVING = 0.5, VIDEO_OVER_AUDIO = 1.25;
const int VIDEO_DELAY_FLOOR = -15;

/* frames per second that `rate` bytes/s carry at `size` bytes a frame (:368-379) */
uint32_t fps_for(uint64_t size, uint64_t rate)
{
    if (rate == 0) {
        return 0;
    }
    const double ms = (double)size * 1000.0 / (double)rate;
    const double f = ms != 0.0 ? 1000.0 / ms : (double)FPS_CEIL;
    return (uint32_t)f;
}

/* jitter buffer a client needs: two frame send times plus latency (:1094-1108) */
uint32_t playback_delay_for(uint64_t size, uint64_t rate, uint32_t latency)
{
    if (size == 0 || rate == 0) {
        return latency;
    }
    const uint32_t send_ms = (uint32_t)((size * 1000) / rate);
    const uint32_t want = send_ms * 2 + latency;
    return want < PLAYBACK_DELAY_CAP_MS ? want : PLAYBACK_DELAY_CAP_MS;
}

uint64_t monotonic_ns(void *)
{
    struct timespec ts;
    clock_gettime(CLOCK_MONOTONIC, &ts);
    return (uint64_t)ts.tv_sec * 1000000000ull + (uint64_t)ts.tv_nsec;
}

} // namespace

uint64_t MjpegRate::now() const
{
    return clock_ ? clock_(clock_opaque_) : monotonic_ns(nullptr);
}

uint32_t MjpegRate::latency_ms() const
{
    return cbs_.get_roundtrip_ms ? cbs_.get_roundtrip_ms(cbs_opaque_) / 2 : 0; /* :362-366 */
}

uint32_t MjpegRate::source_fps() const
{
    return cbs_.get_source_fps(cbs_opaque_);
}

void MjpegRate::tell_playback_delay(uint32_t ms) const
{
    if (cbs_.update_client_playback_delay) {
        delay_updates_++;
        delay_last_ = ms;
        cbs_.update_client_playback_delay(cbs_opaque_, ms);
    }
}

void MjpegRate::set_clock(ClockFn fn, void *opaque)
{
    clock_ = fn;
    clock_opaque_ = opaque;
    if (active_ && warmup_from_ns_ != 0) {
        warmup_from_ns_ = now(); /* a running warm-up restarts on the new time base */
    }
}

/* server/mjpeg_encoder.c:194-214 */
void MjpegRate::init(bool active, uint64_t starting_bit_rate,
                     const MJpegEncoderRateControlCbs *cbs, void *cbs_opaque)
{
    active_ = active;
    byte_rate_ = starting_bit_rate / 8;
    if (!active) {
        settle(MJPEG_RATE_STATIC_RUNG, FPS_CEIL, 0);
        return;
    }
    cbs_ = *cbs;
    cbs_opaque_ = cbs_opaque;
    settle(MID_RUNG, 5, 0);
    probing_ = true;
    probe_.kind = SET;
    probe_.cause = RATE_CHANGE;
    warmup_from_ns_ = now();
}

/* End any probe and run at (rung, fps); frame_size is the size the choice was
 * based on (server/mjpeg_encoder.c:381-418). */
void MjpegRate::settle(int rung, uint32_t fps, uint64_t frame_size)
{
    probing_ = false;
    if (rung_ != rung) {
        last_size_ = 0;
    }
    if (probe_.cause == RATE_CHANGE) {
        coded_ = dropped_ = 0;
    }
    rung_ = rung;
    memset(&probe_, 0, sizeof(probe_));
    probe_.ceil_rung = MJPEG_RATE_LADDER - 1;
    probe_.ceil_fps = (int)FPS_CEIL;

    /* keep the ratio pacing had reached over the nominal rate */
    const double pace_ratio = paced_fps_ != 0.0 ? paced_fps_ / (double)fps_ : 1.5;
    fps_ = fps < FPS_FLOOR ? FPS_FLOOR : fps;
    fps_ = fps_ > FPS_CEIL ? FPS_CEIL : fps_;
    paced_fps_ = (double)fps_ * pace_ratio;
    paced_since_ns_ = 0;
    paced_frames_ = 0;
    settled_size_ = frame_size;
    recent_bytes_ = 0;
    recent_frames_ = 0;
}

void MjpegRate::begin_upgrade(int cause, int floor_rung, int floor_fps)
{
    probing_ = true; /* :572-584 */
    probe_.kind = UPGRADE;
    probe_.cause = cause;
    probe_.floor_rung = floor_rung;
    probe_.floor_fps = floor_fps;
}

void MjpegRate::begin_downgrade(int cause, int ceil_rung, int ceil_fps)
{
    probing_ = true; /* :586-598 */
    probe_.kind = DOWNGRADE;
    probe_.cause = cause;
    probe_.ceil_rung = ceil_rung;
    probe_.ceil_fps = ceil_fps;
}

/* A probe interrupted by a budget change falls back to the bound it started
 * from (server/mjpeg_encoder.c:886-916). */
void MjpegRate::cancel_probe()
{
    if (!probing_) {
        return;
    }
    switch (probe_.kind) {
    case UPGRADE:
        settle(probe_.floor_rung, (uint32_t)probe_.floor_fps, 0);
        break;
    case DOWNGRADE:
        settle(probe_.ceil_rung, (uint32_t)probe_.ceil_fps, 0);
        break;
    case SET:
        settle(MID_RUNG, FPS_CEIL / 2, 0);
        break;
    default:
        break;
    }
}

/* One step of the ladder probe, run before a frame is coded: judge the rung the
 * previous frame was coded on, then move to a neighbouring rung or finish
 * (server/mjpeg_encoder.c:439-535). */
void MjpegRate::probe_step()
{
    const uint64_t size = probe_.size_at[rung_];
    if (size == 0) {
        return;
    }
    const uint32_t src = source_fps();
    const uint32_t fps = fps_for(size, byte_rate_);

    if (fps > (uint32_t)probe_.best_fps ||
        ((fps == (uint32_t)probe_.best_fps || fps >= src) && rung_ > probe_.best_fps_rung)) {
        probe_.best_fps = (int)fps;
        probe_.best_fps_rung = rung_;
    }

    const bool seen_below = rung_ > 0 && probe_.size_at[rung_ - 1] != 0;
    const bool slow = fps < FPS_STRICT && fps < src;

    if (rung_ > MID_RUNG && slow) {
        /* above the middle rung a usable frame rate matters more than quality */
        rung_--;
        if (seen_below) {
            finish_probe();
        }
        return;
    }
    if ((fps > FPS_LENIENT && (double)fps >= 0.66 * (double)probe_.floor_fps) || fps >= src) {
        const bool top = rung_ + 1 == MJPEG_RATE_LADDER || rung_ >= probe_.ceil_rung ||
                         probe_.size_at[rung_ + 1] != 0;
        if (top || (rung_ == MID_RUNG && slow)) {
            finish_probe();
        } else {
            rung_++;
        }
        return;
    }
    /* hardly any frame rate: step down if there is room */
    if (rung_ == 0 || rung_ <= probe_.floor_rung) {
        finish_probe();
        return;
    }
    rung_--;
    if (seen_below) {
        finish_probe();
    }
}

/* server/mjpeg_encoder.c:537-570 */
void MjpegRate::finish_probe()
{
    int rung = rung_;
    if (probe_.best_fps != 0 && probe_.best_fps_rung > rung) {
        rung = probe_.best_fps_rung; /* a better rung gave the better rate after all */
    }
    const uint64_t size = probe_.size_at[rung];
    uint32_t fps = fps_for(size, byte_rate_);
    if (rung == probe_.floor_rung && fps < (uint32_t)probe_.floor_fps) {
        fps = (uint32_t)probe_.floor_fps;
    }
    if (rung == probe_.ceil_rung && fps > (uint32_t)probe_.ceil_fps) {
        fps = (uint32_t)probe_.ceil_fps;
    }
    settle(rung, fps, size);
    recent_bytes_ = size;
    recent_frames_ = 1;
    if (cbs_.update_client_playback_delay) {
        tell_playback_delay(playback_delay_for(size, byte_rate_, latency_ms()));
    }
}

/* Before a frame: feed the last frame's size to a running probe, or see whether
 * recent sizes moved the affordable fps enough to start one
 * (server/mjpeg_encoder.c:600-679). */
void MjpegRate::track_frame_size()
{
    if (!active_ || last_size_ == 0) {
        return;
    }
    if (probing_) {
        probe_.size_at[rung_] = last_size_;
        probe_step();
        return;
    }
    uint64_t avg = 0;
    if (recent_frames_ >= SIZE_WINDOW || recent_frames_ >= fps_) {
        (void)latency_ms(); /* the reference samples the round trip here (:634) */
        avg = recent_bytes_ / recent_frames_;
        const uint32_t can = fps_for(avg, byte_rate_);
        const uint32_t src = source_fps();
        if (can > fps_ && (fps_ < src || rung_ < MJPEG_RATE_LADDER - 1)) {
            begin_upgrade(SIZE_CHANGE, rung_, (int)fps_); /* never below where it stands */
        } else if (can < fps_ && can < src) {
            begin_downgrade(SIZE_CHANGE, rung_, (int)fps_);
        }
    }
    if (probing_) {
        probe_.size_at[rung_] = avg;
        probe_step();
    } else {
        weigh_server_drops();
    }
}

/* Frames do not arrive evenly, so the admitted rate is corrected from the
 * measured average every half second (server/mjpeg_encoder.c:688-727). */
void MjpegRate::retune_fps(uint64_t t)
{
    if (!active_) {
        return;
    }
    const uint64_t ms = (t - paced_since_ns_) / 1000 / 1000;
    if (probing_ || !(ms > PACE_REVIEW_MS) || !((double)ms > 1000.0 / paced_fps_)) {
        return;
    }
    const double avg = ((double)paced_frames_ * 1000.0) / (double)ms;
    const double ratio = avg / (double)fps_;
    if (avg + 0.5 < (double)fps_ && (double)source_fps() > avg) {
        const double want = avg != 0.0 ? paced_fps_ / ratio : paced_fps_ * 2;
        const double cap = (double)(fps_ * 2);
        paced_fps_ = cap < want ? cap : want;
    } else if ((double)fps_ + 0.5 < avg) {
        const double want = paced_fps_ / ratio;
        paced_fps_ = (double)fps_ > want ? (double)fps_ : want;
    }
    paced_since_ns_ = t;
    paced_frames_ = 0;
}

/* server/mjpeg_encoder.c:736-769 */
bool MjpegRate::admit_frame(uint32_t frame_mm_time)
{
    if (!active_) {
        return true;
    }
    const uint64_t t = now();
    if (paced_since_ns_ == 0) {
        paced_since_ns_ = t;
    }
    retune_fps(t);
    const uint64_t gap = t - budget_.last_frame_ns;
    if ((double)gap < 1e9 / paced_fps_) {
        return false;
    }
    track_frame_size();
    if (!probing_ || probe_.cause == SIZE_CHANGE) {
        if (budget_.since_ns == 0) {
            budget_.since_ns = t;
            budget_.since_mm = frame_mm_time;
        }
        budget_.last_frame_ns = t;
    }
    return true;
}

/* server/mjpeg_encoder.c:865-882 */
void MjpegRate::frame_coded(uint64_t size)
{
    last_size_ = size;
    coded_++;
    if (probing_ && probe_.cause != SIZE_CHANGE) {
        return;
    }
    if (!probing_) {
        if (recent_frames_ >= SIZE_WINDOW) {
            recent_frames_ = 0;
            recent_bytes_ = 0;
        }
        recent_bytes_ += size;
        recent_frames_++;
        paced_frames_++;
    }
    budget_.bytes += size;
    budget_.frames++;
}

void MjpegRate::forget_budget_history(int raised)
{
    budget_.since_ns = 0;
    budget_.since_mm = 0;
    budget_.last_frame_ns = 0;
    budget_.frames = 0;
    budget_.bytes = 0;
    budget_.raised = raised;
}

/* server/mjpeg_encoder.c:918-984 */
void MjpegRate::lower_budget()
{
    cancel_probe();
    peak_video_delay_ = 0;
    peak_audio_delay_ = 0;
    if (warmup_from_ns_ != 0) {
        /* drops right after a stream starts say little about the bit rate */
        if (now() - warmup_from_ns_ < WARMUP_NS) {
            return;
        }
        warmup_from_ns_ = 0;
    }
    uint64_t measured;
    uint64_t cut;
    if (budget_.frames > BUDGET_MIN_FRAMES || budget_.frames > fps_) {
        const double secs = (double)(budget_.last_frame_ns - budget_.since_ns) / (1000.0 * 1000.0 * 1000.0);
        measured = (uint64_t)((double)budget_.bytes / secs);
        cut = budget_.bytes / budget_.frames;
    } else {
        measured = byte_rate_;
        cut = measured / fps_;
    }
    measured = byte_rate_ < measured ? byte_rate_ : measured;
    if (cut >= measured) {
        cut = measured / 2;
    }
    byte_rate_ = measured - cut;
    forget_budget_history(0);
    begin_downgrade(RATE_CHANGE, rung_, (int)fps_);
}

/* server/mjpeg_encoder.c:1003-1055 */
void MjpegRate::raise_budget()
{
    if (!(budget_.frames > BUDGET_MIN_FRAMES || budget_.frames > fps_)) {
        return; /* too few frames to tell what the link carries */
    }
    const double secs = (double)(budget_.last_frame_ns - budget_.since_ns) / (1000.0 * 1000.0 * 1000.0);
    const uint64_t measured = (uint64_t)((double)budget_.bytes / secs);
    const uint64_t step = budget_.bytes / budget_.frames;

    cancel_probe();
    if (!(measured + step < byte_rate_)) {
        byte_rate_ = (measured < byte_rate_ ? measured : byte_rate_) + step;
    }
    forget_budget_history(1);
    begin_upgrade(RATE_CHANGE, rung_, (int)fps_);
}

/* server/mjpeg_encoder.c:986-1001 */
void MjpegRate::negative_report(uint32_t end_mm_time)
{
    if ((budget_.since_mm > end_mm_time || budget_.since_mm == 0) && !budget_.raised) {
        return; /* the budget was already lowered after the time the report covers */
    }
    lower_budget();
}

/* server/mjpeg_encoder.c:1056-1088 */
void MjpegRate::positive_report(uint32_t start_mm_time)
{
    if (probing_ && probe_.cause == RATE_CHANGE) {
        return;
    }
    const bool comfortable = (fps_ > FPS_STRICT || fps_ >= source_fps()) && rung_ > MID_RUNG;
    const int wait = comfortable ? REPORT_WAIT_STRICT_MM : REPORT_WAIT_MM;
    const int stable_for = (int)(start_mm_time - budget_.since_mm);
    if (budget_.since_mm == 0 || stable_for < wait) {
        return; /* the present budget has not been in force long enough to judge */
    }
    raise_budget();
}

/* server/mjpeg_encoder.c:1114-1220 */
void MjpegRate::client_report(uint32_t num_frames, uint32_t num_drops, uint32_t start_mm_time,
                              uint32_t end_mm_time, int32_t end_frame_delay, uint32_t audio_delay)
{
    (void)num_frames;
    if (!active_) {
        return;
    }
    if (probing_ && probe_.kind == DOWNGRADE && probe_.cause == RATE_CHANGE) {
        return;
    }
    const uint64_t avg = recent_frames_ ? recent_bytes_ / recent_frames_ : 0;
    const uint32_t need_delay = playback_delay_for(avg, byte_rate_, latency_ms());
    bool client_delay_short = false;

    /* the reference compares the two as unsigned: a negative delay never counts as short here */
    if (need_delay > (uint32_t)end_frame_delay) {
        const uint32_t src = source_fps();
        const uint32_t full = src < FPS_CEIL ? src : FPS_CEIL;
        /* at the top rung and full rate the real link rate cannot be told */
        if (rung_ != MJPEG_RATE_LADDER - 1 || fps_ < full || end_frame_delay < 0) {
            client_delay_short = true;
            tell_playback_delay(need_delay);
        }
    }

    /* audio latency fell while video latency stayed high: video is too fast */
    if (end_frame_delay > 0 &&
        (double)audio_delay < LATENCY_HALVING * (double)peak_audio_delay_ &&
        (double)end_frame_delay > VIDEO_OVER_AUDIO * (double)audio_delay) {
        negative_report(end_mm_time);
        return;
    }
    if (end_frame_delay < VIDEO_DELAY_FLOOR) {
        negative_report(end_mm_time);
        return;
    }
    if (end_frame_delay > peak_video_delay_) {
        peak_video_delay_ = end_frame_delay;
    }
    if (audio_delay > peak_audio_delay_) {
        peak_audio_delay_ = audio_delay;
    }
    const double half_peak = (double)peak_video_delay_ * LATENCY_HALVING;
    const double quarter_peak = half_peak * LATENCY_HALVING;
    if (((double)end_frame_delay < half_peak && client_delay_short) ||
        (double)end_frame_delay < quarter_peak) {
        negative_report(end_mm_time);
    } else if (num_drops == 0) {
        positive_report(start_mm_time);
    }
}

/* server/mjpeg_encoder.c:1222-1258 */
void MjpegRate::server_dropped_frame()
{
    dropped_++;
    weigh_server_drops();
}

void MjpegRate::weigh_server_drops()
{
    const uint32_t src = source_fps();
    const uint32_t per_sec = fps_ < src ? fps_ : src;
    if (coded_ < per_sec) {
        return; /* judged once a second's worth of frames was coded */
    }
    const uint32_t total = dropped_ + coded_;
    const double share = (double)dropped_ / (double)total;
    if (share > SERVER_DROP_SHARE) {
        lower_budget();
    }
    coded_ = 0;
    dropped_ = 0;
}

void MjpegRate::snapshot(SpiceGpuMjpegRateState *s) const
{
    memset(s, 0, sizeof(*s));
    s->during_quality_eval = probing_;
    s->eval_type = probe_.kind;
    s->eval_reason = probe_.cause;
    s->min_quality_id = probe_.floor_rung;
    s->min_quality_fps = probe_.floor_fps;
    s->max_quality_id = probe_.ceil_rung;
    s->max_quality_fps = probe_.ceil_fps;
    s->max_sampled_fps = probe_.best_fps;
    s->max_sampled_fps_quality_id = probe_.best_fps_rung;
    for (int i = 0; i < MJPEG_RATE_LADDER; i++) {
        s->encoded_size_by_quality[i] = probe_.size_at[i];
    }
    s->quality_id = rung_;
    s->fps = fps_;
    s->adjusted_fps = paced_fps_;
    s->adjusted_fps_start_time = paced_since_ns_;
    s->adjusted_fps_num_frames = paced_frames_;
    s->byte_rate = byte_rate_;
    s->base_enc_size = settled_size_;
    s->last_enc_size = last_size_;
    s->sum_recent_enc_size = recent_bytes_;
    s->num_recent_enc_frames = recent_frames_;
    s->warmup_start_time = warmup_from_ns_;
    s->change_start_time = budget_.since_ns;
    s->last_frame_time = budget_.last_frame_ns;
    s->change_start_mm_time = budget_.since_mm;
    s->was_upgraded = budget_.raised;
    s->num_enc_frames = budget_.frames;
    s->sum_enc_size = budget_.bytes;
    s->max_video_latency = peak_video_delay_;
    s->max_audio_latency = peak_audio_delay_;
    s->num_frames_encoded = coded_;
    s->num_frames_dropped = dropped_;
    s->playback_delay_updates = delay_updates_;
    s->playback_delay_last = delay_last_;
}

} // namespace spicegpu
