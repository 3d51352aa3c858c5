"""crt_image_export_fd: the frame handed to another API as a dma-buf descriptor instead of being read back.

There is no Vulkan in this image, so the importer is HIP itself: the descriptor is imported the way any external-memory
consumer does (import, reserve, map), through a second mapping that shares nothing with the library's own but the
physical pages, and what it shows must be the image crt_image_read returns, bit for bit."""
import os
import subprocess
import sys

import pytest

from computeraytracer_amd import abi

pytestmark = pytest.mark.gpu


def test_exported_descriptor_shows_the_frame():
    # in a fresh interpreter (tests/export_check.py says why)
    here = os.path.dirname(os.path.abspath(__file__))
    env = dict(os.environ, PYTHONPATH=os.pathsep.join([os.path.dirname(here), here, os.environ.get("PYTHONPATH", "")]))
    r = subprocess.run([sys.executable, os.path.join(here, "export_check.py")], env=env, capture_output=True, text=True, timeout=240)
    assert r.returncode == 0 and "export-ok" in r.stdout, r.stdout[-2000:] + r.stderr[-4000:]


def test_export_needs_a_configured_context():
    with abi.Context(0) as c:
        with pytest.raises(abi.CrtError, match="not configured"):
            c.image_export_fd()
