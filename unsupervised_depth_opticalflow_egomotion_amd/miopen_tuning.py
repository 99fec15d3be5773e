"""Tuned MIOpen databases for the step's convolution problems (MI355X, the MIOpen build of the ROCm 7.2 image).

The networks' convolutions are MIOpen calls (convs.py).  For every problem MIOpen picks a solver from its find-db and
kernel parameters from its perf-db; where the shipped system databases have no entry for gfx950 it falls back on
heuristics.  ``miopen_db/`` holds the *user* databases of one auto-tuning pass over the training step's problems
(``MIOPEN_FIND_ENFORCE=3 python bench.py --steps 2``, 754 s on one MI355X, tools/miopen_tune.sh): text files keyed by
architecture, CU count and MIOpen version -- on any other machine or MIOpen build they are simply not read.  Measured:
26.00 -> 25.64 ms per step (profiles/r03_miopen_tuned_db.txt); the first step also skips most of the find phase.

``activate()`` copies them into a per-user scratch directory (MIOpen appends what it learns about new problems to its user
databases; the tracked files are never written) and points ``MIOPEN_USER_DB_PATH`` there -- before the first
convolution, and only if the variable is not set already.  ``DFE_MIOPEN_DB=0`` leaves MIOpen alone."""
import hashlib
import os
import shutil
import tempfile

DB_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "miopen_db")
_state = {"path": None, "done": False}


def activate():
    """-> the user-db directory in use (None: MIOpen's own default)."""
    if _state["done"]:
        return _state["path"]
    _state["done"] = True
    if os.environ.get("MIOPEN_USER_DB_PATH"):
        _state["path"] = os.environ["MIOPEN_USER_DB_PATH"]
        return _state["path"]
    if os.environ.get("DFE_MIOPEN_DB", "1") == "0" or not os.path.isdir(DB_DIR):
        return None
    files = sorted(f for f in os.listdir(DB_DIR) if f.endswith(".txt"))
    if not files:
        return None
    h = hashlib.sha256()
    for f in files:
        with open(os.path.join(DB_DIR, f), "rb") as fh:
            h.update(f.encode()); h.update(fh.read())
    dst = os.path.join(tempfile.gettempdir(), "dfe_miopen_db_%d_%s" % (os.getuid(), h.hexdigest()[:10]))
    try:
        os.makedirs(dst, exist_ok=True)
        for f in files:
            target = os.path.join(dst, f)
            if not os.path.exists(target):          # several ranks may do this at once: copy aside, then rename
                tmp = "%s.%d.tmp" % (target, os.getpid())
                shutil.copyfile(os.path.join(DB_DIR, f), tmp)
                os.replace(tmp, target)
    except OSError:
        return None
    os.environ["MIOPEN_USER_DB_PATH"] = dst
    _state["path"] = dst
    return dst


def status():
    """'tuned' when the shipped databases are the ones MIOpen reads, 'env' for a caller-chosen path, else 'default'."""
    p = _state["path"]
    if p is None:
        return "default"
    return "tuned" if os.path.basename(p).startswith("dfe_miopen_db_") else "env"
