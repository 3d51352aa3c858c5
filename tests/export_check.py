"""The body of tests/test_gpu_export.py::test_exported_descriptor_shows_the_frame, run as a program in a fresh interpreter:
the importer must use the HIP runtime libcrt.so itself is linked against.  Inside the pytest process another test may have
imported torch first, whose bundled (older) runtime then answers to the same soname, and the two runtimes disagree about
hipMemImportFromShareableHandle's osHandle argument (the descriptor itself in ROCm 7.2, a pointer to it in the older one)."""
import os

import numpy as np

from computeraytracer_amd import abi
from hip_util import Hip


def _render(c, scene, w, h, n, first=1):
    seeds = abi.frame_seeds(first - 1, n)
    c.render(scene.params, first, seeds)


def main():
    hip = Hip()
    w, h = 200, 112
    scene = abi.builtin_scene("book1", w, h)
    with abi.Context(0) as c:
        c.configure(w, h)
        c.upload(scene); c.build()
        _render(c, scene, w, h, 2)
        fd, nbytes, abytes = c.image_export_fd()
        assert fd >= 0 and nbytes == w * h * 16 and abytes >= nbytes
        assert os.lseek(fd, 0, os.SEEK_END) >= nbytes        # a dma-buf reports its size
        c.sync()
        want = c.image()
        va, handle = hip.import_fd(fd, abytes)
        got = np.zeros((h, w, 4), np.float32)
        hip.d2h(got, va)
        assert np.array_equal(got.view(np.uint32), want.view(np.uint32)) and (got[..., 3] == 2).all()
        # the importer keeps its mapping: later frames arrive through it with no new descriptor
        _render(c, scene, w, h, 1, first=3)
        fd2, nb2, ab2 = c.image_export_fd(want_fd=False)
        assert fd2 is None and (nb2, ab2) == (nbytes, abytes)
        c.sync()
        hip.d2h(got, va)
        assert np.array_equal(got.view(np.uint32), c.image().view(np.uint32)) and (got[..., 3] == 3).all()
        hip.release_import(va, handle, abytes)
        os.close(fd)
        # a different frame size gets a new allocation; a band-sharded context exports the rows it owns
        c.configure(96, 64, band_rank=1, band_world=2)
        c.upload(abi.builtin_scene("quads", 96, 64)); c.build()
        _render(c, abi.builtin_scene("quads", 96, 64), 96, 64, 1)
        fd3, nb3, ab3 = c.image_export_fd()
        assert nb3 == 4 * 8 * 96 * 16          # bands 1, 3, 5, 7 of eight
        c.sync()
        va3, h3 = hip.import_fd(fd3, ab3)
        part = np.zeros((32, 96, 4), np.float32)
        hip.d2h(part, va3)
        full = c.image()
        assert np.array_equal(part.view(np.uint32), full[abi.band_rows(64, 1, 2)].view(np.uint32))
        hip.release_import(va3, h3, ab3)
        os.close(fd3)



if __name__ == "__main__":
    main()
    print("export-ok")
