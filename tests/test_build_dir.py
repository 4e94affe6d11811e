"""tools/build_dir.py: shared objects built in the scratch directory are ctypes-loaded and its generated headers are compiled into
the product, so a directory somebody else prepared (shared, a symlink, another owner) must not be used."""
import os
import stat
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import build_dir as bd  # noqa: E402


def test_a_shared_or_linked_scratch_directory_is_replaced_by_a_private_one(tmp_path, monkeypatch):
    monkeypatch.setattr(bd, "_FALLBACK", None)
    shared = tmp_path / "shared"
    shared.mkdir()
    os.chmod(shared, 0o777)
    monkeypatch.setenv("ROBOY_BUILD_DIR", str(shared))
    got = bd.build_dir()
    st = os.lstat(got)
    assert got != str(shared) and stat.S_ISDIR(st.st_mode) and st.st_uid == os.getuid() and (st.st_mode & 0o077) == 0
    link = tmp_path / "link"
    os.symlink(got, link)
    monkeypatch.setenv("ROBOY_BUILD_DIR", str(link))
    assert bd.build_dir() == got                      # (the one private fallback of this process, not the link)
    own = tmp_path / "own"
    monkeypatch.setenv("ROBOY_BUILD_DIR", str(own))
    assert bd.build_dir() == str(own) and (os.lstat(own).st_mode & 0o077) == 0
