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
import mmap
import os
import re
import shutil
import stat
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
        os.makedirs(dst, mode=0o700, exist_ok=True)
        st = os.stat(dst)                           # a predictable name under /tmp: somebody else may have made it
        if st.st_uid != os.getuid() or stat.S_IMODE(st.st_mode) & 0o022:
            return None
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


def shipped_keys():
    """The database keys of the shipped files: '<arch><CUs in hex>.HIP.<MIOpen version tag>'."""
    if not os.path.isdir(DB_DIR):
        return set()
    return {re.sub(r"\.(udb|ufdb)\.txt$", "", f) for f in os.listdir(DB_DIR) if f.endswith(".txt")}


def running_key():
    """The key the MIOpen in this process reads its user databases under, or None when it cannot be established
    (no device, no libMIOpen next to torch).  Architecture and CU count come from the device properties; the version
    tag ('3_5_0_20250912-42-1199-g2584e35062') is the build string embedded in libMIOpen.so."""
    try:
        import torch
        if not torch.cuda.is_available():
            return None
        prop = torch.cuda.get_device_properties(torch.cuda.current_device())
        arch = prop.gcnArchName.split(":")[0]
        lib = os.path.join(os.path.dirname(torch.__file__), "lib", "libMIOpen.so")
        with open(lib, "rb") as fh:
            mm = mmap.mmap(fh.fileno(), 0, access=mmap.ACCESS_READ)
            m = re.search(rb"(\d+)\.(\d+)\.(\d+)\.(\d{8}-[0-9A-Za-z\-]{3,40})\x00", mm)
            parts = None if m is None else tuple(x.decode() for x in m.groups())
            del m                                   # the match holds the mapping's buffer
            mm.close()
        if parts is None:
            return None
        tag = "%s_%s_%s_%s" % parts
        return "%s%x.HIP.%s" % (arch, prop.multi_processor_count, tag)
    except Exception:
        return None


def status():
    """'tuned' when the shipped databases are the ones MIOpen reads AND one of them carries the running MIOpen's key;
    'tuned-unmatched' when they are in place but keyed for another architecture / CU count / MIOpen build (MIOpen then
    simply does not open them); 'tuned-unverified' when the running key cannot be established; 'env' for a
    caller-chosen path, else 'default'."""
    p = _state["path"]
    if p is None:
        return "default"
    if not os.path.basename(p).startswith("dfe_miopen_db_"):
        return "env"
    key = running_key()
    if key is None:
        return "tuned-unverified"
    return "tuned" if key in shipped_keys() else "tuned-unmatched"
