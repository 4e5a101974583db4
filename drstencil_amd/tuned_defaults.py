"""tuned_defaults -- the tuner -> generator feedback table (reference: the flow of benchmarks/3d7pt_star/tuning.py:125-131 ends with
the best configuration in duration.log, which a user then copies into the command line by hand).

`tuned_defaults.tsv` holds one row per tuned problem class:

    mode  shape  points  order  step  dtype  temporal  N  |  generator options  |  source (the tuner log the row came from)

mode = 3d | 2d (one-shot tile) | 2ds (--streaming); shape = FNV-1a hash of the ONE-STEP stencil's sorted offsets (two stencils with
the same number of points need not want the same kernel: star and cross), points / order = its point count and order (outermost-dim
reach; for the reader, the lookup goes by shape); N = the innermost grid size the search ran on.  `tuning.py --write-defaults` appends / replaces rows from a search's
results; `write_header()` turns the table into csrc/tuned_defaults.hpp, which the generator consults when a command line names a
problem (--3d / --dtype / --step / --streaming / --temporal and the .stc) but no geometry or emission option: the row of the same
class whose N is nearest in log2 (within a factor of sqrt 2) supplies them.  bench.py builds its TUNED option lists from the same
rows.  tests/test_cli_and_ir.py checks that the header is in sync and that the bare command line emits the tuned kernel."""
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
TABLE = os.path.join(_HERE, "tuned_defaults.tsv")
HEADER = os.path.join(_HERE, "csrc", "tuned_defaults.hpp")
FIELDS = ("mode", "shape", "step", "dtype", "temporal", "N")


def load(path=TABLE):
    rows = []
    for line in open(path):
        line = line.rstrip("\n")
        if not line.strip() or line.lstrip().startswith("#"):
            continue
        key, opts, src = [x.strip() for x in line.split("|")]
        mode, shape, points, order, step, dtype, temporal, n = key.split()
        rows.append(dict(mode=mode, shape=shape, points=int(points), order=int(order), step=int(step), dtype=dtype, temporal=int(temporal), N=int(n),
                         options=opts, source=src))
    return rows


def shape_hash(offsets):
    """FNV-1a (32 bit) over the offsets in lexicographic (k, j, i) order, as "k,j,i;" text; the C++ twin is tuned_shape_hash()."""
    h = 0x811c9dc5
    for k, j, i in sorted(offsets):
        for ch in ("%d,%d,%d;" % (k, j, i)).encode():
            h = ((h ^ ch) * 0x01000193) & 0xffffffff
    return "%08x" % h


def lookup(mode, shape, step, dtype, temporal, N, rows=None):
    """Options (list) of the nearest row of this class, or None.  The C++ twin is tuned_lookup() in csrc/generator.hpp."""
    import math
    best = None
    for r in rows if rows is not None else load():
        if (r["mode"], r["shape"], r["step"], r["dtype"], r["temporal"]) != (mode, shape, step, dtype, int(temporal)):
            continue
        d = abs(math.log2(N / r["N"]))
        if d <= 0.5 + 1e-9 and (best is None or d < best[0]):
            best = (d, r)
    return best[1]["options"].split() if best else None


def read_stc(spec_path, ndim):
    """The .stc as the generator reads it (stencil_ir.hpp read_stc; reference drstencil.hpp:52-78), in plain Python -- bench.py's launcher
    process builds its option lists from the table and must not load the native library: (dims dict, sorted offsets as (k, j, i))."""
    toks = open(spec_path).read().split()
    dims, offs, i = {}, set(), 0
    while i < len(toks):
        t = toks[i]
        if t in ("L", "M", "N", "iterations") and i + 1 < len(toks) and not (t == "L" and ndim != 3):
            dims[t] = int(toks[i + 1])
            i += 2
        elif t == "stencil":
            w = ndim + 1
            body = toks[i + 1:]
            for r in range(len(body) // w):
                try:
                    o = tuple(int(x) for x in body[r * w:r * w + ndim])
                    float(body[r * w + ndim])
                except ValueError:
                    break
                offs.add(o if ndim == 3 else (0,) + o)
            break
        else:
            i += 1
    return dims, sorted(offs)


def key_of(spec_path, ndim, streaming=False):
    """(mode, shape hash, points, order) of the one-step stencil in `spec_path` and its innermost size N."""
    dims, offs = read_stc(spec_path, ndim)
    mode = "3d" if ndim == 3 else ("2ds" if streaming else "2d")
    order = max([abs(o[0] if ndim == 3 else o[1]) for o in offs] or [0])
    return mode, shape_hash(offs), len(offs), order, dims.get("N", 0)


def problem_options(ndim, dtype, step=1, temporal=0, streaming=False, dist=None):
    """The options that NAME a problem (everything else is tuning)."""
    o = (["--3d"] if ndim == 3 else []) + ["--dtype", dtype]
    if step != 1:
        o += ["--step", str(step)]
    if dist:
        o += ["--dist", str(dist)]
    if temporal:
        o += ["--temporal", "1"]
    if streaming:
        o += ["--streaming"]
    return o


def options_for(spec_path, ndim, dtype, step=1, temporal=0, streaming=False, dist=None, rows=None):
    """Full generator option list for a problem: the naming options + the table row's (KeyError when the class has no row)."""
    mode, shape, points, order, N = key_of(spec_path, ndim, streaming)
    t = lookup(mode, shape, step, dtype, temporal, N, rows)
    if t is None:
        raise KeyError("no tuned defaults for %s %s (%d points, order %d) step %d %s temporal %d N %d" % (mode, shape, points, order, step, dtype, temporal, N))
    return problem_options(ndim, dtype, step, temporal, streaming, dist) + t


def put(row, path=TABLE):
    """Insert or replace the row of `row`'s class and N; rewrites the table (sorted) and the header."""
    rows = [r for r in load(path) if tuple(r[f] for f in FIELDS) != tuple(row[f] for f in FIELDS)] if os.path.exists(path) else []
    rows.append(row)
    save(rows, path)


def save(rows, path=TABLE):
    rows = sorted(rows, key=_order)
    with open(path, "w") as f:
        f.write("# tuner -> generator feedback table (drstencil_amd/tuned_defaults.py; written by `tuning.py --write-defaults`)\n")
        f.write("# mode shape points order step dtype temporal N | generator options | source\n")
        for r in rows:
            f.write("%s %s %d %d %d %s %d %d | %s | %s\n" % (r["mode"], r["shape"], r["points"], r["order"], r["step"], r["dtype"], r["temporal"], r["N"], r["options"], r["source"]))
    if path == TABLE:
        write_header(rows)


def _order(r):
    return (r["mode"], r["points"], r["order"], r["shape"], r["dtype"], r["step"], r["temporal"], r["N"])


def header_text(rows):
    esc = lambda s: s.replace("\\", "\\\\").replace('"', '\\"')
    out = ["// tuned_defaults.hpp -- GENERATED from drstencil_amd/tuned_defaults.tsv by drstencil_amd/tuned_defaults.py (tuning.py --write-defaults): do not edit.",
           "// The tuner's winners per problem class; generator.hpp applies a row when a command line gives no geometry / emission option.",
           "#pragma once", "namespace drs {",
           "struct TunedDefault { const char *mode; unsigned shape; int points, order, step; const char *dtype; int temporal, N; const char *options; };",
           "static const TunedDefault kTunedDefaults[] = {"]
    for r in sorted(rows, key=_order):
        out.append('    {"%s", 0x%su, %d, %d, %d, "%s", %d, %d, "%s"},' % (r["mode"], r["shape"], r["points"], r["order"], r["step"], r["dtype"], r["temporal"], r["N"], esc(r["options"])))
    out += ["};", "static const int kTunedDefaultsCount = (int)(sizeof(kTunedDefaults) / sizeof(kTunedDefaults[0]));", "}  // namespace drs", ""]
    return "\n".join(out)


def write_header(rows=None, path=HEADER):
    with open(path, "w") as f:
        f.write(header_text(load() if rows is None else rows))


if __name__ == "__main__":
    write_header()
    print("wrote", HEADER)
