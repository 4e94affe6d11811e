"""Where the tests' and generators' scratch builds go (g++ builds of the product's host code, generated headers,
assembly listings): a per-user directory OUTSIDE the repository, so that none of it travels to the GPU box with a
`gpurun` snapshot (round 4 shipped 77 MB of it per call).  ROBOY_BUILD_DIR overrides.

Shared objects built there are loaded with ctypes and the generator's output is compiled into libroboy_sim.so, so the
directory must be this user's own and closed to everybody else: a predictable name under /tmp that somebody else created
(or may write to) is not used - a fresh private directory is made instead."""
import os
import stat
import tempfile

_FALLBACK = None


def _private(d):
    """d exists, is a real directory (not a symlink), belongs to this user and nobody else may write or read it"""
    try:
        st = os.lstat(d)
    except OSError:
        return False
    return stat.S_ISDIR(st.st_mode) and st.st_uid == os.getuid() and (st.st_mode & 0o077) == 0


def build_dir():
    global _FALLBACK
    d = os.environ.get("ROBOY_BUILD_DIR") or os.path.join(tempfile.gettempdir(), "roboy_amd_build_%d" % os.getuid())
    try:
        os.makedirs(d, mode=0o700, exist_ok=True)
    except OSError:
        pass
    if _private(d):
        return d
    if _FALLBACK is None:          # planted, shared or unwritable: a directory nobody could have prepared (one per process)
        _FALLBACK = tempfile.mkdtemp(prefix="roboy_amd_build_")
    return _FALLBACK
