"""CPU: static sanity of the MATLAB side (there is no MATLAB / Octave here to run it): every .m file under matlab/ and tools/ has
balanced block keywords, a leading function line whose name matches the file, and calls into the MEX gateway only with commands
the gateway implements."""
import glob
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
MFILES = sorted(glob.glob(os.path.join(ROOT, "matlab", "*.m")) + glob.glob(os.path.join(ROOT, "tools", "*.m")))


def strip(txt):
    out = []
    for ln in txt.split("\n"):
        ln = re.sub(r"'[^']*'", "''", ln)
        ln = re.sub(r'"[^"]*"', '""', ln)
        out.append(ln.split("%")[0])
    return "\n".join(out)


def test_block_keywords_balance():
    assert MFILES
    for f in MFILES:
        code = strip(open(f).read())
        opens = len(re.findall(r"(?<![\w.])(function|if|for|while|switch|try|parfor)(?![\w])", code))
        ends = len(re.findall(r"(?<![\w.])end(?![\w(])", code))
        index_like = len(re.findall(r"[\(\[,:]\s*end\b|end\s*[\)\],:+\-]", code))      # x(end), x(end-1), a:end
        assert opens == ends or opens == ends - index_like, (os.path.basename(f), opens, ends, index_like)


def test_function_name_matches_file():
    for f in MFILES:
        first = next(ln for ln in open(f).read().split("\n") if ln.strip() and not ln.strip().startswith("%"))
        m = re.match(r"\s*function\s+(?:\[[^\]]*\]\s*=\s*|\w+\s*=\s*)?(\w+)", first)
        assert m and m.group(1) == os.path.splitext(os.path.basename(f))[0], (os.path.basename(f), first)


def test_mex_commands_exist_in_the_gateway():
    gateway = open(os.path.join(ROOT, "matlab", "vbmc_hip_mex.cpp")).read()
    implemented = set(re.findall(r'cmd\s*==\s*"(\w+)"', gateway)) | set(re.findall(r'!strcmp\(cmd,\s*"(\w+)"\)', gateway))
    assert implemented, "no commands recognised in the gateway source"
    used = set()
    for f in MFILES:
        used |= set(re.findall(r"vbmc_hip_mex\(\s*'(\w+)'", open(f).read()))
    assert used and used <= implemented, sorted(used - implemented)


def _signature(path):
    txt = open(path, errors="replace").read()
    m = re.search(r"^\s*function\s+(?:\[([^\]]*)\]|(\w+))?\s*=?\s*(\w+)\s*\(([^)]*)\)", txt, re.M)
    assert m, path
    outs = [t.strip() for t in re.split(r"[,\s]+", (m.group(1) or m.group(2) or "").strip()) if t.strip()]
    args = [t.strip() for t in m.group(4).split(",") if t.strip()]
    return m.group(3), outs, args


def test_shim_signatures_match_the_reference():
    """Drop-in by path lookup only works if a same-named shim takes and returns what the reference function does: compare the
    function lines (names and ORDER of inputs and outputs) with the reference's files where the reference is available (this
    container; skipped on a box without /root/reference)."""
    import pytest

    ref_root = "/root/reference"
    if not os.path.isdir(ref_root):
        pytest.skip("reference not present")
    index = {}
    for dp, _, fn in os.walk(ref_root):
        for f in fn:
            if f.endswith(".m"):
                index.setdefault(f, []).append(os.path.join(dp, f))
    checked = 0
    for f in sorted(glob.glob(os.path.join(ROOT, "matlab", "*.m"))):
        base = os.path.basename(f)
        if base.startswith("vbmc_hip_") or base not in index:
            continue                      # helpers of the shim layer itself
        # gplite/private/sq_dist.m and utils/sq_dist.m are the same function; any one of the candidates must match
        name, outs, args = _signature(f)
        cands = [_signature(p) for p in index[base]]
        assert any(name == n and len(outs) == len(o) and [a.lower() for a in args] == [b.lower() for b in a_]
                   for n, o, a_ in cands), (base, outs, args, cands)
        checked += 1
    assert checked >= 8
