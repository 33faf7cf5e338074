"""A second, independent statement of the GLZ window's bookkeeping, for checking
what the library reports (spice_gpu_glz_dictionary_dump) after every encode.

It is written from what server/glz_encoder_dictionary.c documents and does, on
plain Python lists -- no linked lists, no shared text with the library's host
code or with the oracle:

  * images leave the window oldest first, and only as many as the next image
    needs: the window's size is taken as the segments' pixel counts from the
    head's first segment to the tail's last, and it shrinks by an image's size
    for every image that leaves (glz_encoder_dictionary.c:392-424);
  * an image's id is the count of images the dictionary has ever been given
    (:492); a segment's pixels_so_far is the sum of the pixel counts of all
    segments before it, ever (:472-473); a chunk's pixel count is
    image_size * rows / height in integers (:506);
  * segment slots: 1000 at first, handed out in index order; the slots of an
    image that leaves go to the FRONT of the free list, in the image's own order
    (:370-387); when the list is empty the table doubles and the new slots are
    handed out in index order (:267-309);
  * free_image fires for every image that leaves alive, oldest first (:359-367);
  * after an encode has returned no encoder holds the window (:593-633).
"""

NULL_SEG = 0xFFFFFFFF
BPP = {6: 2, 7: 3, 8: 4, 9: 4}


class WindowModel:
    def __init__(self, size_limit, max_encoders=1, last_image_id=0):
        self.limit, self.max_encoders = size_limit, max_encoders
        self.images = []                 # oldest first: dict(id, size, alive, tag, type, segs=[(slot, psf, n)])
        self.free = list(range(1000))
        self.slots = 1000
        self.next_id = last_image_id
        self.psf = 0
        self.freed = []                  # tags handed to free_image since the last pop

    def _window_pixels(self):
        if not self.images:
            return 0
        first = self.images[0]["segs"][0]
        last = self.images[-1]["segs"][-1]
        return last[1] + last[2] - first[1]

    def _leave(self):
        im = self.images.pop(0)
        self.free = [s[0] for s in im["segs"]] + self.free
        if im["alive"]:
            self.freed.append(im["tag"])

    def add(self, img_type, height, stride, chunk_rows, tag):
        size = height * stride // BPP[img_type]
        assert size <= self.limit
        win = self._window_pixels()
        while self.images and win + size > self.limit:
            win -= self.images[0]["size"]
            self._leave()
        segs, rows = [], 0
        for r in chunk_rows:
            if not self.free:
                self.free = list(range(self.slots, 2 * self.slots))
                self.slots *= 2
            n = size * r // height
            segs.append((self.free.pop(0), self.psf, n))
            self.psf += n
            rows += r
            if rows >= height:
                break
        self.images.append(dict(id=self.next_id, size=size, alive=True, tag=tag, type=img_type, segs=segs))
        self.next_id += 1

    def remove(self, tag):
        for im in self.images:
            if im["tag"] == tag:
                im["alive"] = False

    def reset(self):
        while self.images:
            im = self.images.pop(0)
            if im["alive"]:
                self.freed.append(im["tag"])
        self.free = list(range(self.slots))
        self.next_id = 0

    def pop_freed(self):
        f, self.freed = self.freed, []
        return f

    def check(self, dump, what=""):
        """dump: GlzDictionary.dump() taken when no encode is running"""
        assert dump["size_limit"] == self.limit and dump["max_encoders"] == self.max_encoders, what
        assert dump["last_image_id"] == self.next_id, what
        assert dump["seg_slots"] == self.slots, what
        assert all(h == NULL_SEG for h in dump["encoders_heads"]), what + ": an encoder still holds the window"
        got = dump["images"]
        assert [im["id"] for im in got] == [im["id"] for im in self.images], what + ": images in the window"
        psf0 = None
        for g, m in zip(got, self.images):
            assert (g["size"], g["alive"], g["usr_context"], g["type"]) == \
                (m["size"], m["alive"], m["tag"], m["type"]), what + ": image %d" % m["id"]
            assert [s["slot"] for s in g["segs"]] == [s[0] for s in m["segs"]], what + ": slots of image %d" % m["id"]
            assert [s["pixels_num"] for s in g["segs"]] == [s[2] for s in m["segs"]], what
            for gs, ms in zip(g["segs"], m["segs"]):
                if psf0 is None:            # the dictionary's counter may have started elsewhere (a reset keeps it)
                    psf0 = gs["pixels_so_far"] - ms[1]
                assert gs["pixels_so_far"] - ms[1] == psf0, what + ": pixels_so_far"
        # the used segments form one chain in window order, closed by the null slot
        chain = [s for g in got for s in g["segs"]]
        for a, b in zip(chain, chain[1:]):
            assert a["next"] == b["slot"], what + ": segment chain"
        if chain:
            assert chain[-1]["next"] == NULL_SEG and dump["used_segs_head"] == chain[0]["slot"] and \
                dump["used_segs_tail"] == chain[-1]["slot"], what
            win = chain[-1]["pixels_so_far"] + chain[-1]["pixels_num"] - chain[0]["pixels_so_far"]
            assert win <= self.limit, what + ": %d pixels in a window of %d" % (win, self.limit)
        else:
            assert dump["used_segs_head"] == NULL_SEG and dump["used_segs_tail"] == NULL_SEG, what
        assert dump["free_segs"] == self.free, what + ": order of the free segment slots"
        used = [s["slot"] for s in chain]
        assert len(set(used + dump["free_segs"])) == len(used) + len(dump["free_segs"]) == self.slots, what
