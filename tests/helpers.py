"""Shared pieces of the parity tests."""
import numpy as np

from spice_amd import binding as B
from spice_amd import frames

from window_model import WindowModel


def pixels_for(typ, bgrx, rng):
    """Rows (h, stride) of `bgrx` in the pixel layout of LzImageType `typ`."""
    h, w = bgrx.shape[:2]
    if typ == B.LZ_IMAGE_TYPE_RGB24:
        return frames.to_bgr24(bgrx)
    if typ == B.LZ_IMAGE_TYPE_RGB16:
        return frames.to_rgb16(bgrx, rng)
    a = bgrx.copy()
    if typ == B.LZ_IMAGE_TYPE_RGBA:
        a[..., 3] = (np.arange(w)[None, :] // 8 + np.arange(h)[:, None] // 4) & 0xff
    else:
        a[..., 3] = rng.integers(0, 256, (h, w))  # pad must be ignored by RGB32
    return a.reshape(h, w * 4)


def split_rows(px, nchunks):
    h = px.shape[0]
    parts = np.array_split(np.arange(h), max(1, min(nchunks, h)))
    return [np.ascontiguousarray(px[p[0]:p[-1] + 1]) for p in parts if len(p)]


def small_desktop(w, h, seed):
    seq = frames.DesktopSequence(max(w, 96), max(h, 96), seed)
    return seq.next()[:h, :w].copy()


def assert_same_stream(expected, got, what=""):
    if len(expected) != len(got) or not (expected == got).all():
        n = min(len(expected), len(got))
        diff = np.nonzero(expected[:n] != got[:n])[0]
        first = int(diff[0]) if len(diff) else n
        raise AssertionError("%s: streams differ (len %d vs %d, first difference at byte %d)"
                             % (what, len(expected), len(got), first))


class GlzPair:
    """One oracle dictionary and one device dictionary fed the same images."""

    def __init__(self, gpu, oracle, size_px, max_encoders=1, batch=False):
        """batch=True sends the images through the device-resident batch interface
        (pixels uploaded first, stream downloaded) instead of glz_encode"""
        self.gpu, self.O, self.batch = gpu, oracle, batch
        self.od = oracle.GlzDict(size_px, max_encoders)
        self.gd = B.GlzDictionary(gpu, size_px, max_encoders)
        self.ge = B.GlzEncoder(gpu, self.gd, 0)
        self.model = WindowModel(size_px, max_encoders)      # independent bookkeeping, see window_model.py
        self.n = 0

    def remove(self, oimg, gimg, tag):
        """glz_enc_dictionary_remove_image on both sides"""
        self.od.remove_image(oimg)
        self.gd.remove_image(gimg)
        self.model.remove(tag)

    def encode(self, typ, w, h, stride, chunks, top_down=True, what="", direct=False):
        tag = 1000 + self.n
        self.n += 1
        so, oimg = self.od.encode(typ, w, h, stride, chunks, top_down=top_down, usr_tag=tag)
        if self.batch:
            dev = self.gpu
            held = [dev.to_device(c) for c in chunks]
            cap = dev.L.spice_gpu_glz_out_bound(typ, w, h)
            out = dev.malloc(cap)
            try:
                sizes, imgs = B.glz_encode_batch(dev, [dict(
                    encoder=self.ge, type=typ, width=w, height=h, stride=stride, top_down=top_down,
                    chunks=held, chunk_rows=[c.shape[0] for c in chunks], out=out, out_capacity=cap,
                    usr_context=tag)])
                sg, gimg = dev.download(out, sizes[0]), imgs[0]
            finally:
                dev.free(out)
                for p in held:
                    dev.free(p)
        else:
            sg, gimg = self.ge.encode(typ, w, h, stride, chunks, top_down=top_down, usr_context=tag,
                                      first_lines_direct=direct)
        assert_same_stream(so, sg, "%s image %d (%dx%d type %d)" % (what, self.n - 1, w, h, typ))
        assert self.od.pop_freed() == self.gd.usr.freed_images
        self.model.add(typ, h, stride, [c.shape[0] for c in chunks], tag)
        assert self.model.pop_freed() == self.gd.usr.freed_images, what
        self.model.check(self.gd.dump(), "%s image %d" % (what, self.n - 1))
        self.gd.usr.freed_images.clear()
        return so, oimg, gimg

    def close(self):
        self.ge.close()
        self.gd.close()
        self.od.close()
