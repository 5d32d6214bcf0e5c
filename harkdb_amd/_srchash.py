"""Does the in-tree libhark.so come from the sources in the tree?  The Makefile writes the sha256 of the concatenated sources
next to the library when it links it (harkdb_amd/libhark.srchash); this module computes the same digest from the tree.
bench.py reports the comparison, tools/evidence.sh refuses to measure when it fails (a library left behind by an experimental
build of edited sources once produced an evidence run: profiles/r04_notes.md)."""
import glob
import hashlib
import os

HERE = os.path.dirname(os.path.abspath(__file__))


def sources_digest():
    csrc = os.path.join(HERE, "csrc")
    files = sorted(glob.glob(os.path.join(csrc, "*.hip")), key=os.path.basename)
    files += [os.path.join(csrc, "hark_internal.h"), os.path.join(csrc, "sort_networks.h"),
              os.path.join(HERE, "..", "include", "hark.h"), os.path.join(HERE, "..", "include", "futhark_compat.h")]
    h = hashlib.sha256()
    for f in files:
        with open(f, "rb") as fh:
            h.update(fh.read())
    return h.hexdigest()


def library_matches_sources():
    """True / False, or None when there is no record (a library built by other means)."""
    try:
        with open(os.path.join(HERE, "libhark.srchash")) as fh:
            return fh.read().strip() == sources_digest()
    except OSError:
        return None


if __name__ == "__main__":
    import sys
    ok = library_matches_sources()
    print({True: "libhark.so was linked from the sources in the tree", False: "libhark.so was NOT linked from the sources in the tree: run make -C harkdb_amd/csrc",
           None: "no harkdb_amd/libhark.srchash: run make -C harkdb_amd/csrc"}[ok])
    sys.exit(0 if ok else 1)
