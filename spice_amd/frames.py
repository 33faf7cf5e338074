"""Seeded synthetic inputs for the compression path (numpy only, no I/O).

plumbing   the update bitmaps of the reference's display test harness
           (server/tests/test_display_base.c:212-271): W/4 x 48 BGRX tiles from
           its per-pixel formula, and solid fills
desktop    desktop-like 32-bit frames: flat background, overlapping windows
           with 1-pixel borders, two-colour text cells, a gradient; successive
           frames scroll one window by 16 rows and blink a caret, so that GLZ
           finds cross-image matches
video      a smooth sinusoid field with a moving textured block and +-2 LSB
           noise, the kind of content the stream detector hands to MJPEG
sessions64 64 independent display sessions, every one a desktop sequence with
           its own seed; the even ones are the GLZ half, the odd ones the half
           whose bitmaps score a HIGH graduality and go to QUIC in the reference
multimon   one client with four 3840x2160 monitors: per tick every monitor
           sends a desktop-like update of the upper 70 % of its area (GLZ, all
           four through the client's one shared dictionary) and a video frame
           of the lower 30 % (MJPEG)
"""
import numpy as np


def plumbing_tile(primary_width, t, color=0, c_i=0):
    """test_spice_create_update_draw (test_display_base.c:255-266); pad byte zeroed
    (the harness leaves it uninitialised)."""
    bw, bh = primary_width // 4, 48
    i = np.arange(bw * bh, dtype=np.int64)
    px = np.zeros((bw * bh, 4), dtype=np.uint8)
    px[:, 0] = (color + i % 255) & 0xff
    px[:, (1 + c_i) % 3] = (255 - color) & 0xff
    px[:, (2 + c_i) % 3] = (color * (color + i)) & 0xff
    px[:, (3 + c_i) % 3] = 0
    return px.reshape(bh, bw, 4)


def plumbing_solid(width, height, color):
    """test_spice_create_update_solid (test_display_base.c:212-231): 0x00RRGGBB words."""
    a = np.empty((height, width), dtype=np.uint32)
    a[:] = color
    return a.view(np.uint8).reshape(height, width, 4)


def _glyphs(rng, n=96):
    g = rng.integers(0, 2, size=(n, 16, 8), dtype=np.uint8)
    g[:, :3, :] = 0
    g[:, 13:, :] = 0
    g[:, :, 7] = 0
    return g


class DesktopSequence:
    """Frames of one desktop session; frame k+1 = frame k with small edits."""

    def __init__(self, width=1920, height=1080, seed=1):
        self.w, self.h = width, height
        rng = self.rng = np.random.default_rng(seed)
        self.glyphs = _glyphs(rng)
        f = np.zeros((height, width, 4), dtype=np.uint8)
        f[..., :3] = rng.integers(30, 200, 3)
        nwin = int(rng.integers(6, 11))
        self.windows = []
        for _ in range(nwin):
            ww = int(rng.integers(max(32, width // 8), max(40, width // 2)))
            wh = int(rng.integers(max(32, height // 8), max(40, height // 2)))
            x0 = int(rng.integers(0, max(1, width - ww)))
            y0 = int(rng.integers(0, max(1, height - wh)))
            self.windows.append((x0, y0, ww, wh))
            self._draw_window(f, x0, y0, ww, wh)
        gs = min(256, width // 2, height // 2)
        gx, gy = int(rng.integers(0, width - gs + 1)), int(rng.integers(0, height - gs + 1))
        yy, xx = np.mgrid[0:gs, 0:gs]
        f[gy:gy + gs, gx:gx + gs, 0] = (xx * 255 // max(1, gs - 1)).astype(np.uint8)
        f[gy:gy + gs, gx:gx + gs, 1] = (yy * 255 // max(1, gs - 1)).astype(np.uint8)
        f[gy:gy + gs, gx:gx + gs, 2] = ((xx + yy) * 255 // max(1, 2 * gs - 2)).astype(np.uint8)
        self.frame = f
        self.scroll_win = self.windows[-1]
        self.k = 0

    def _draw_window(self, f, x0, y0, ww, wh):
        rng = self.rng
        bg = rng.integers(200, 256, 3).astype(np.uint8)
        fg = rng.integers(0, 80, 3).astype(np.uint8)
        f[y0:y0 + wh, x0:x0 + ww, :3] = bg
        border = rng.integers(0, 256, 3).astype(np.uint8)
        f[y0, x0:x0 + ww, :3] = border
        f[y0 + wh - 1, x0:x0 + ww, :3] = border
        f[y0:y0 + wh, x0, :3] = border
        f[y0:y0 + wh, x0 + ww - 1, :3] = border
        self._text(f, x0 + 4, y0 + 4, (ww - 8) // 8, (wh - 8) // 16, bg, fg)

    def _text(self, f, x0, y0, cols, rows, bg, fg):
        if cols <= 0 or rows <= 0:
            return
        idx = self.rng.integers(0, len(self.glyphs), size=(rows, cols))
        spaces = self.rng.random((rows, cols)) < 0.18
        cells = self.glyphs[idx]
        cells[spaces] = 0
        bitmap = cells.transpose(0, 2, 1, 3).reshape(rows * 16, cols * 8)
        region = f[y0:y0 + rows * 16, x0:x0 + cols * 8, :3]
        region[:] = np.where(bitmap[..., None] != 0, fg, bg)

    def next(self):
        """Returns the next frame (a fresh array, BGRX, pad 0)."""
        if self.k > 0:
            f = self.frame
            x0, y0, ww, wh = self.scroll_win
            ix0, iy0, iw, ih = x0 + 1, y0 + 1, ww - 2, wh - 2
            if ih > 32:
                inner = f[iy0:iy0 + ih, ix0:ix0 + iw].copy()
                f[iy0:iy0 + ih - 16, ix0:ix0 + iw] = inner[16:]
                bg = inner[0, 0, :3].copy()
                fg = np.array([20, 20, 20], dtype=np.uint8)
                f[iy0 + ih - 16:iy0 + ih, ix0:ix0 + iw, :3] = bg
                self._text(f, ix0 + 3, iy0 + ih - 16, (iw - 6) // 8, 1, bg, fg)
            cx, cy, cw, ch = self.windows[0]
            caret = f[cy + 6:cy + 20, cx + 6:cx + 8, :3]
            caret[:] = 255 - caret
        self.k += 1
        return self.frame.copy()


class VideoSequence:
    def __init__(self, width=3840, height=2160, seed=7):
        self.w, self.h = width, height
        self.rng = np.random.default_rng(seed)
        y, x = np.mgrid[0:height, 0:width].astype(np.float32)
        self.x, self.y = x, y
        bs = min(512, width // 2, height // 2)
        self.bs = bs
        self.block = self.rng.integers(0, 256, size=(bs, bs, 3), dtype=np.uint8)
        self.k = 0

    def next(self):
        k, w, h = self.k, self.w, self.h
        ph = 0.07 * k
        f = np.empty((h, w, 4), dtype=np.uint8)
        sx = 2 * np.pi / max(w, 1)
        sy = 2 * np.pi / max(h, 1)
        base = [127 + 100 * np.sin(self.x * sx * 1.5 + ph) * np.cos(self.y * sy + 0.5 * ph),
                127 + 90 * np.sin(self.x * sx + self.y * sy * 2 + 1.3 * ph),
                127 + 80 * np.cos(self.x * sx * 0.7 - self.y * sy * 1.1 + 0.8 * ph)]
        noise = self.rng.integers(-2, 3, size=(h, w, 3))
        for c in range(3):
            f[..., c] = np.clip(base[c] + noise[..., c], 0, 255).astype(np.uint8)
        f[..., 3] = 0
        bs = self.bs
        bx = (37 * k) % max(1, w - bs)
        by = (23 * k) % max(1, h - bs)
        f[by:by + bs, bx:bx + bs, :3] = self.block
        self.k += 1
        return f


def to_rgb16(bgrx, rng=None):
    """x1r5g5b5 little-endian words from a BGRX frame (top bit random if rng)."""
    b = bgrx[..., 0].astype(np.uint16) >> 3
    g = bgrx[..., 1].astype(np.uint16) >> 3
    r = bgrx[..., 2].astype(np.uint16) >> 3
    v = b | (g << 5) | (r << 10)
    if rng is not None:
        v = v | (rng.integers(0, 2, size=v.shape).astype(np.uint16) << 15)
    h, w = v.shape
    return v.view(np.uint8).reshape(h, w * 2)


def to_bgr24(bgrx):
    return np.ascontiguousarray(bgrx[..., :3]).reshape(bgrx.shape[0], -1)


def hash_noise(n, seed):
    """n deterministic pseudo-random bytes (splitmix64 of the index); independent
    of any library's generator so that committed golden vectors stay valid."""
    with np.errstate(over="ignore"):
        z = (np.arange(n, dtype=np.uint64) + np.uint64(seed) * np.uint64(0x9E3779B97F4A7C15))
        z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
        z = z ^ (z >> np.uint64(31))
    return (z & np.uint64(0xff)).astype(np.uint8)


def golden_image(width, height, kind, seed):
    """BGRX test images used by the committed golden vectors."""
    if kind == "noise":
        a = hash_noise(width * height * 4, seed).reshape(height, width, 4).copy()
    elif kind == "smooth":
        y, x = np.mgrid[0:height, 0:width]
        a = np.stack([(x * 3 + y + seed) % 256, (x + y * 2) % 256, (x * y // 7) % 256,
                      np.zeros_like(x)], -1).astype(np.uint8)
        a[..., :3] ^= (hash_noise(width * height * 3, seed).reshape(height, width, 3) & 3)
    elif kind == "desktop":
        a = DesktopSequence(max(width, 96), max(height, 96), seed).next()[:height, :width].copy()
    else:
        raise ValueError(kind)
    a[..., 3] = 0
    return a


class Sessions64:
    """BASELINE configs[3]: 64 concurrent display sessions, each a desktop sequence of
    its own.  In the reference the sessions whose updates are smooth (graduality HIGH)
    are coded with QUIC and the others with GLZ (server/red_worker.c:6462-6505); here
    the odd-numbered sessions stand for the QUIC half and the even ones for the GLZ half."""

    N = 64

    def __init__(self, width=1920, height=1080, seed=640):
        self.w, self.h, self.seed = width, height, seed

    def session_ids(self, codec):
        return [c for c in range(self.N) if (c % 2 == 0) == (codec == "glz")]

    def sequence(self, session_id):
        """the session's frames: DesktopSequence for the GLZ half; the QUIC half gets
        smooth gradients (what scores HIGH)"""
        if session_id % 2 == 0:
            return DesktopSequence(self.w, self.h, seed=self.seed + session_id)
        return _SmoothSequence(self.w, self.h, seed=self.seed + session_id)


class _SmoothSequence:
    def __init__(self, width, height, seed):
        self.w, self.h, self.seed, self.k = width, height, seed, 0

    def next(self):
        self.k += 1
        return golden_image(self.w, self.h, "smooth", self.seed + self.k)


class MultimonClient:
    """BASELINE configs[4]: one client, `monitors` displays of width x height.  Every
    tick each monitor yields (desktop region for GLZ, video region for MJPEG): the
    upper 70 % of the rows and the lower 30 %.  The regions are independent
    sequences, so a monitor's GLZ updates scroll and blink like a desktop and its
    video region moves like a video."""

    def __init__(self, width=3840, height=2160, monitors=4, seed=4000):
        self.w, self.h, self.monitors = width, height, monitors
        self.glz_rows = height * 7 // 10
        self.video_rows = height - self.glz_rows
        self.desk = [DesktopSequence(width, self.glz_rows, seed=seed + 10 * m) for m in range(monitors)]
        self.video = [VideoSequence(width, self.video_rows, seed=seed + 10 * m + 5) for m in range(monitors)]

    def tick(self):
        """[(glz region (rows, w, 4), video region (rows, w, 4)) per monitor]"""
        return [(d.next(), v.next()) for d, v in zip(self.desk, self.video)]
