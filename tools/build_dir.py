"""Where the tests' and generators' scratch builds go (g++ builds of the product's host code, generated headers,
assembly listings): a per-user directory OUTSIDE the repository, so that none of it travels to the GPU box with a
`gpurun` snapshot (round 4 shipped 77 MB of it per call).  ROBOY_BUILD_DIR overrides."""
import os
import tempfile


def build_dir():
    d = os.environ.get("ROBOY_BUILD_DIR") or os.path.join(tempfile.gettempdir(), "roboy_amd_build_%d" % os.getuid())
    os.makedirs(d, mode=0o700, exist_ok=True)
    return d
