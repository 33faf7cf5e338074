/*
 * TEST INFRASTRUCTURE ONLY -- CPU restatement of the reference's MJPEG stream
 * rate control.  Not linked into the product; only tests/ use it.
 *
 * Follows server/mjpeg_encoder.c: mjpeg_encoder_new (:187-220),
 * get_max_fps (:368-379), mjpeg_encoder_reset_quality (:381-418),
 * mjpeg_encoder_eval_quality (:439-570), the upgrade/downgrade setters
 * (:572-598), mjpeg_encoder_adjust_params_to_bit_rate (:600-679),
 * mjpeg_encoder_adjust_fps (:688-727), the rate part of start_frame
 * (:736-769) and end_frame (:865-882), quality_eval_stop (:886-916),
 * decrease/increase_bit_rate (:918-1055), the client report handlers
 * (:986-1001, :1056-1088, :1114-1220) and the server drop handlers
 * (:1222-1258) -- same order of tests, same C integer/double conversions.
 * The clock (CLOCK_MONOTONIC in the reference) and the two callbacks'
 * return values are inputs here.  mjpeg_encoder.c cannot be compiled in this
 * image (red_common.h pulls in the absent spice-protocol / spice-common
 * headers) and the reference holds no test for it: parity unpinned.
 *
 * Where the reference would abort on spice_assert(num_recent_enc_frames)
 * (:627) the restatement sets `asserted` and stops changing state.
 */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define NQ 7
#define MAXF 25
#define MINF 1
#define HI(a, b) ((a) > (b) ? (a) : (b))
#define LO(a, b) ((a) < (b) ? (a) : (b))

enum { T_SET, T_UP, T_DOWN };
enum { R_SIZE, R_RATE };

/* same layout as SpiceGpuMjpegRateState (include/spice_gpu.h) so one ctypes
 * class reads both */
typedef struct {
    int32_t during_quality_eval, eval_type, eval_reason;
    int32_t min_quality_id, min_quality_fps, max_quality_id, max_quality_fps;
    int32_t max_sampled_fps, max_sampled_fps_quality_id;
    int32_t quality_id;
    uint32_t fps;
    uint32_t num_recent_enc_frames;
    uint64_t encoded_size_by_quality[NQ];
    double adjusted_fps;
    uint64_t adjusted_fps_start_time, adjusted_fps_num_frames;
    uint64_t byte_rate, base_enc_size, last_enc_size, sum_recent_enc_size;
    uint64_t warmup_start_time;
    uint64_t change_start_time, last_frame_time, sum_enc_size;
    uint32_t change_start_mm_time, num_enc_frames;
    int32_t was_upgraded, max_video_latency;
    uint32_t max_audio_latency;
    uint32_t num_frames_encoded, num_frames_dropped;
    uint32_t playback_delay_updates, playback_delay_last;
    uint32_t reserved;
} MjroState;

typedef struct {
    MjroState s;
    uint64_t now;          /* the clock, set by the caller before each event */
    uint32_t roundtrip_ms; /* what get_roundtrip_ms returns */
    uint32_t source_fps;   /* what get_source_fps returns */
    int has_delay_cb;
    int asserted;
    uint64_t sum_quality;
    uint32_t num_frames;
} Mjro;

static const int ladder[NQ] = {20, 30, 40, 50, 60, 70, 80};

static void server_drops(Mjro *m);

static uint32_t latency(Mjro *m) { return m->roundtrip_ms / 2; }

static uint32_t max_fps(uint64_t frame_size, uint64_t bytes_per_sec)
{
    double fps, send_ms;
    if (!bytes_per_sec) {
        return 0;
    }
    send_ms = frame_size * 1000.0 / bytes_per_sec;
    fps = send_ms ? 1000 / send_ms : MAXF;
    return fps;
}

static uint32_t min_playback_delay(uint64_t size, uint64_t byte_rate, uint32_t lat)
{
    uint32_t one, d;
    if (!size || !byte_rate) {
        return lat;
    }
    one = (size * 1000) / byte_rate;
    d = LO(one * 2 + lat, 5000);
    return d;
}

static void delay_cb(Mjro *m, uint32_t ms)
{
    if (m->has_delay_cb) {
        m->s.playback_delay_updates++;
        m->s.playback_delay_last = ms;
    }
}

static void reset_quality(Mjro *m, int quality_id, uint32_t fps, uint64_t size)
{
    MjroState *s = &m->s;
    double ratio;
    s->during_quality_eval = 0;
    if (s->quality_id != quality_id) {
        s->last_enc_size = 0;
    }
    if (s->eval_reason == R_RATE) {
        s->num_frames_encoded = s->num_frames_dropped = 0;
    }
    s->quality_id = quality_id;
    s->eval_type = s->eval_reason = 0;
    memset(s->encoded_size_by_quality, 0, sizeof(s->encoded_size_by_quality));
    s->min_quality_id = s->min_quality_fps = 0;
    s->max_sampled_fps = s->max_sampled_fps_quality_id = 0;
    s->max_quality_id = NQ - 1;
    s->max_quality_fps = MAXF;
    if (s->adjusted_fps) {
        ratio = s->adjusted_fps / s->fps;
    } else {
        ratio = 1.5;
    }
    s->fps = HI(MINF, fps);
    s->fps = LO(MAXF, s->fps);
    s->adjusted_fps = s->fps * ratio;
    s->adjusted_fps_start_time = 0;
    s->adjusted_fps_num_frames = 0;
    s->base_enc_size = size;
    s->sum_recent_enc_size = 0;
    s->num_recent_enc_frames = 0;
}

#define SEEN(s, q) ((s)->encoded_size_by_quality[(q)] != 0)

static void eval_quality(Mjro *m)
{
    MjroState *s = &m->s;
    uint32_t fps, src, final_id, final_fps;
    uint64_t size, final_size;

    size = s->encoded_size_by_quality[s->quality_id];
    if (size == 0) {
        return;
    }
    src = m->source_fps;
    fps = max_fps(size, s->byte_rate);
    if (fps > (uint32_t)s->max_sampled_fps ||
        ((fps == (uint32_t)s->max_sampled_fps || fps >= src) &&
         s->quality_id > s->max_sampled_fps_quality_id)) {
        s->max_sampled_fps = fps;
        s->max_sampled_fps_quality_id = s->quality_id;
    }
    if (s->quality_id > NQ / 2 && fps < 10 && fps < src) {
        if (SEEN(s, s->quality_id - 1)) {
            s->quality_id--;
            goto complete;
        }
        s->quality_id--;
    } else if ((fps > 5 && fps >= 0.66 * s->min_quality_fps) || fps >= src) {
        if (s->quality_id + 1 == NQ || s->quality_id >= s->max_quality_id ||
            SEEN(s, s->quality_id + 1)) {
            goto complete;
        }
        if (s->quality_id == NQ / 2 && fps < 10 && fps < src) {
            goto complete;
        }
        s->quality_id++;
    } else {
        if (s->quality_id == 0 || s->quality_id <= s->min_quality_id) {
            goto complete;
        } else if (SEEN(s, s->quality_id - 1)) {
            s->quality_id--;
            goto complete;
        }
        s->quality_id--;
    }
    return;

complete:
    if (s->max_sampled_fps != 0) {
        final_id = HI(s->quality_id, s->max_sampled_fps_quality_id);
    } else {
        final_id = s->quality_id;
    }
    final_size = s->encoded_size_by_quality[final_id];
    final_fps = max_fps(final_size, s->byte_rate);
    if (final_id == (uint32_t)s->min_quality_id) {
        final_fps = HI(final_fps, (uint32_t)s->min_quality_fps);
    }
    if (final_id == (uint32_t)s->max_quality_id) {
        final_fps = LO(final_fps, (uint32_t)s->max_quality_fps);
    }
    reset_quality(m, final_id, final_fps, final_size);
    s->sum_recent_enc_size = final_size;
    s->num_recent_enc_frames = 1;
    delay_cb(m, min_playback_delay(final_size, s->byte_rate, latency(m)));
}

static void set_upgrade(Mjro *m, int reason, uint32_t min_id, uint32_t min_fps)
{
    m->s.during_quality_eval = 1;
    m->s.eval_type = T_UP;
    m->s.eval_reason = reason;
    m->s.min_quality_id = min_id;
    m->s.min_quality_fps = min_fps;
}

static void set_downgrade(Mjro *m, int reason, uint32_t max_id, uint32_t max_fps_)
{
    m->s.during_quality_eval = 1;
    m->s.eval_type = T_DOWN;
    m->s.eval_reason = reason;
    m->s.max_quality_id = max_id;
    m->s.max_quality_fps = max_fps_;
}

static void adjust_params(Mjro *m)
{
    MjroState *s = &m->s;
    uint64_t new_avg = 0;
    uint32_t new_fps, src;

    if (!s->last_enc_size) {
        return;
    }
    if (s->during_quality_eval) {
        s->encoded_size_by_quality[s->quality_id] = s->last_enc_size;
        eval_quality(m);
        return;
    }
    if (!s->num_recent_enc_frames) {
        m->asserted = 1;
        return;
    }
    if (s->num_recent_enc_frames < 3 && s->num_recent_enc_frames < s->fps) {
        goto end;
    }
    new_avg = s->sum_recent_enc_size / s->num_recent_enc_frames;
    new_fps = max_fps(new_avg, s->byte_rate);
    src = m->source_fps;
    if (new_fps > s->fps && (s->fps < src || s->quality_id < NQ - 1)) {
        set_upgrade(m, R_SIZE, s->quality_id, s->fps);
    } else if (new_fps < s->fps && new_fps < src) {
        set_downgrade(m, R_SIZE, s->quality_id, s->fps);
    }
end:
    if (s->during_quality_eval) {
        s->encoded_size_by_quality[s->quality_id] = new_avg;
        eval_quality(m);
    } else {
        server_drops(m);
    }
}

static void adjust_fps(Mjro *m, uint64_t now)
{
    MjroState *s = &m->s;
    uint64_t passed = (now - s->adjusted_fps_start_time) / 1000 / 1000;

    if (!s->during_quality_eval && passed > 500 && passed > 1000 / s->adjusted_fps) {
        double avg_fps = ((double)s->adjusted_fps_num_frames * 1000) / passed;
        double ratio = avg_fps / s->fps;
        if (avg_fps + 0.5 < s->fps && m->source_fps > avg_fps) {
            double n = avg_fps ? (s->adjusted_fps / ratio) : s->adjusted_fps * 2;
            s->adjusted_fps = LO(s->fps * 2, n);
        } else if (s->fps + 0.5 < avg_fps) {
            double n = s->adjusted_fps / ratio;
            s->adjusted_fps = HI(s->fps, n);
        }
        s->adjusted_fps_start_time = now;
        s->adjusted_fps_num_frames = 0;
    }
}

static void eval_stop(Mjro *m)
{
    MjroState *s = &m->s;
    if (!s->during_quality_eval) {
        return;
    }
    switch (s->eval_type) {
    case T_UP:
        reset_quality(m, s->min_quality_id, s->min_quality_fps, 0);
        break;
    case T_DOWN:
        reset_quality(m, s->max_quality_id, s->max_quality_fps, 0);
        break;
    case T_SET:
        reset_quality(m, NQ / 2, MAXF / 2, 0);
        break;
    }
}

static void clear_bit_rate_info(MjroState *s, int upgraded)
{
    s->change_start_time = 0;
    s->change_start_mm_time = 0;
    s->last_frame_time = 0;
    s->num_enc_frames = 0;
    s->sum_enc_size = 0;
    s->was_upgraded = upgraded;
}

static void decrease_bit_rate(Mjro *m)
{
    MjroState *s = &m->s;
    uint64_t measured, dec;
    uint32_t measured_fps;

    eval_stop(m);
    s->max_video_latency = 0;
    s->max_audio_latency = 0;
    if (s->warmup_start_time) {
        if (m->now - s->warmup_start_time < 3000L * 1000 * 1000) {
            return;
        }
        s->warmup_start_time = 0;
    }
    if (s->num_enc_frames > 3 || s->num_enc_frames > s->fps) {
        double dur = (s->last_frame_time - s->change_start_time);
        dur /= (1000.0 * 1000.0 * 1000.0);
        measured = s->sum_enc_size / dur;
        dec = s->sum_enc_size / s->num_enc_frames;
    } else {
        measured = s->byte_rate;
        measured_fps = s->fps;
        dec = measured / measured_fps;
    }
    measured = LO(s->byte_rate, measured);
    if (dec >= measured) {
        dec = measured / 2;
    }
    s->byte_rate = measured - dec;
    clear_bit_rate_info(s, 0);
    set_downgrade(m, R_RATE, s->quality_id, s->fps);
}

static void negative_report(Mjro *m, uint32_t end_mm)
{
    MjroState *s = &m->s;
    if ((s->change_start_mm_time > end_mm || !s->change_start_mm_time) && !s->was_upgraded) {
        return;
    }
    decrease_bit_rate(m);
}

static void increase_bit_rate(Mjro *m)
{
    MjroState *s = &m->s;
    uint64_t measured, inc;

    if (s->num_enc_frames > 3 || s->num_enc_frames > s->fps) {
        double dur = (s->last_frame_time - s->change_start_time);
        dur /= (1000.0 * 1000.0 * 1000.0);
        measured = s->sum_enc_size / dur;
        inc = s->sum_enc_size / s->num_enc_frames;
    } else {
        return;
    }
    eval_stop(m);
    if (measured + inc < s->byte_rate) {
        /* measured rate is small: only re-evaluate */
    } else {
        s->byte_rate = LO(measured, s->byte_rate) + inc;
    }
    clear_bit_rate_info(s, 1);
    set_upgrade(m, R_RATE, s->quality_id, s->fps);
}

static void positive_report(Mjro *m, uint32_t start_mm)
{
    MjroState *s = &m->s;
    int stable, timeout;

    if (s->during_quality_eval && s->eval_reason == R_RATE) {
        return;
    }
    if ((s->fps > 10 || s->fps >= m->source_fps) && s->quality_id > NQ / 2) {
        timeout = 3000;
    } else {
        timeout = 2000;
    }
    stable = (int)start_mm - s->change_start_mm_time;
    if (!s->change_start_mm_time || stable < timeout) {
        return;
    }
    increase_bit_rate(m);
}

static void server_drops(Mjro *m)
{
    MjroState *s = &m->s;
    uint32_t total, fps;
    double factor;

    fps = LO(s->fps, m->source_fps);
    if (s->num_frames_encoded < fps * 1) {
        return;
    }
    total = s->num_frames_dropped + s->num_frames_encoded;
    factor = ((double)s->num_frames_dropped) / total;
    if (factor > 0.1) {
        decrease_bit_rate(m);
    }
    s->num_frames_encoded = 0;
    s->num_frames_dropped = 0;
}

/* ------------------------------------------------------------------ exports */

Mjro *mjro_new(uint64_t starting_bit_rate, uint64_t now, int has_delay_cb)
{
    Mjro *m = calloc(1, sizeof(*m));
    m->has_delay_cb = has_delay_cb;
    m->now = now;
    m->s.byte_rate = starting_bit_rate / 8;
    reset_quality(m, NQ / 2, 5, 0);
    m->s.during_quality_eval = 1;
    m->s.eval_type = T_SET;
    m->s.eval_reason = R_RATE;
    m->s.warmup_start_time = now;
    return m;
}

void mjro_free(Mjro *m) { free(m); }

void mjro_env(Mjro *m, uint64_t now, uint32_t roundtrip_ms, uint32_t source_fps)
{
    m->now = now;
    m->roundtrip_ms = roundtrip_ms;
    m->source_fps = source_fps;
}

/* start_frame's rate part; 0 = drop, 1 = encode (quality in *quality), then
 * end_frame's bookkeeping with the size the frame has at the chosen quality
 * (size_by_quality_id[quality_id]; 0 = a scanline failed, :848-851) */
int mjro_feed(Mjro *m, uint32_t mm_time, const uint64_t *size_by_quality_id, int *quality)
{
    MjroState *s = &m->s;
    uint64_t now = m->now, interval, enc_size;

    if (m->asserted) {
        return -1;
    }
    if (!s->adjusted_fps_start_time) {
        s->adjusted_fps_start_time = now;
    }
    adjust_fps(m, now);
    interval = now - s->last_frame_time;
    if (interval < (1000 * 1000 * 1000) / s->adjusted_fps) {
        return 0;
    }
    adjust_params(m);
    if (m->asserted) {
        return -1;
    }
    if (!s->during_quality_eval || s->eval_reason == R_SIZE) {
        if (!s->change_start_time) {
            s->change_start_time = now;
            s->change_start_mm_time = mm_time;
        }
        s->last_frame_time = now;
    }
    *quality = ladder[s->quality_id];
    m->num_frames++;
    m->sum_quality += ladder[s->quality_id];
    enc_size = size_by_quality_id[s->quality_id];
    if (!enc_size) {
        s->last_enc_size = 0;
        return 1;
    }
    s->last_enc_size = enc_size;
    s->num_frames_encoded++;
    if (!s->during_quality_eval || s->eval_reason == R_SIZE) {
        if (!s->during_quality_eval) {
            if (s->num_recent_enc_frames >= 3) {
                s->num_recent_enc_frames = 0;
                s->sum_recent_enc_size = 0;
            }
            s->sum_recent_enc_size += s->last_enc_size;
            s->num_recent_enc_frames++;
            s->adjusted_fps_num_frames++;
        }
        s->sum_enc_size += s->last_enc_size;
        s->num_enc_frames++;
    }
    return 1;
}

void mjro_client_report(Mjro *m, uint32_t num_frames, uint32_t num_drops, uint32_t start_mm,
                        uint32_t end_mm, int32_t end_frame_delay, uint32_t audio_delay)
{
    MjroState *s = &m->s;
    uint64_t avg = 0;
    uint32_t min_delay;
    int small = 0;

    (void)num_frames;
    if (m->asserted) {
        return;
    }
    if (s->during_quality_eval && s->eval_type == T_DOWN && s->eval_reason == R_RATE) {
        return;
    }
    if (s->num_recent_enc_frames) {
        avg = s->sum_recent_enc_size / s->num_recent_enc_frames;
    }
    min_delay = min_playback_delay(avg, s->byte_rate, latency(m));
    if (min_delay > (uint32_t)end_frame_delay) { /* the reference compares uint32_t with int32_t: unsigned */
        uint32_t src = m->source_fps;
        if (s->quality_id != NQ - 1 || s->fps < LO(src, MAXF) || end_frame_delay < 0) {
            small = 1;
            delay_cb(m, min_delay);
        }
    }
    if (end_frame_delay > 0 && audio_delay < 0.5 * s->max_audio_latency &&
        end_frame_delay > 1.25 * audio_delay) {
        negative_report(m, end_mm);
        return;
    }
    if (end_frame_delay < -15) {
        negative_report(m, end_mm);
    } else {
        double medium, major;
        s->max_video_latency = HI(end_frame_delay, s->max_video_latency);
        s->max_audio_latency = HI(audio_delay, s->max_audio_latency);
        medium = s->max_video_latency;
        medium *= 0.5;
        major = medium;
        major *= 0.5;
        if ((end_frame_delay < medium && small) || end_frame_delay < major) {
            negative_report(m, end_mm);
        } else if (!num_drops) {
            positive_report(m, start_mm);
        }
    }
}

void mjro_server_drop(Mjro *m)
{
    if (m->asserted) {
        return;
    }
    m->s.num_frames_dropped++;
    server_drops(m);
}

void mjro_state(const Mjro *m, MjroState *out) { *out = m->s; }
int mjro_asserted(const Mjro *m) { return m->asserted; }
uint64_t mjro_bit_rate(const Mjro *m) { return m->s.byte_rate * 8; }
double mjro_avg_quality(const Mjro *m) { return (double)m->sum_quality / m->num_frames; }
