"""Randomised differential run of hc_yaml_read (hydrochrono_amd/csrc/hc_yaml.cpp) against the REFERENCE's own hydro.yaml parser
(oracle/_ref/libref_yaml.so = /root/reference/src/hydro_yaml_parser.cpp compiled as it lies): documents drawn from the keys, synonyms,
value spellings, block / inline forms, indentation, comments, quoting, line endings and orderings the format knows -- plus deliberate
damage (missing keys, wrong indentation, duplicated and unknown keys, garbage values) -- must be accepted with the same fields or rejected
with the same message by both.  CPU only.   python profiles/fuzz_yaml.py [cases = 20000] [first seed = 1]"""
import os
import sys
import tempfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import test_hydro_yaml as T  # noqa: E402

ncases = int(sys.argv[1]) if len(sys.argv) > 1 else 20000
seed0 = int(sys.argv[2]) if len(sys.argv) > 2 else 1


def number(rng):
    k = rng.integers(0, 12)
    v = float(rng.choice([0.0, 1.0, 2.5, 7.0, 12.0, 0.75, 1e-3, 42.0, 100.0, 9.5, 0.6, 0.95]))
    return [f"{v}", f"{v:g}", f"{int(v)}", f"{v:.3e}", f"-{v:g}", f"+{v:g}", f"{v:g} ", f"  {v:g}", f"{v:g}  # c", f'"{v:g}"', ".5", "1e3"][k]


def boolean(rng):
    return str(rng.choice(["true", "false", "yes", "no", "True", "False", "on", "off", "1", "0", "TRUE", "y", "n", ""]))


def word(rng, pool):
    w = str(rng.choice(pool))
    q = rng.integers(0, 8)
    return [w, w, w, f'"{w}"', f"'{w}'", w.upper(), w.capitalize(), f"{w} # note"][q]


def body(rng, i, ind):
    pad = " " * ind
    lines = []
    keys = [("name", word(rng, ["float", "spar", "body1", "sphere", "b 2", "flap"])),
            ("h5_file", word(rng, ["x.h5", "../hydroData/rm3.h5", "/abs/p/y.h5", "hydroData/sphere.h5", "a b.h5"]))]
    opt = [("include_excitation", boolean(rng)), ("include_radiation", boolean(rng)),
           ("radiation_calculation", word(rng, ["convolution", "state_space", "none", "conv"])),
           ("radiation_convolution_mode", word(rng, ["Baseline", "TaperedDirect", "baseline", "tapered_direct", "other"])),
           ("td_smoothing", word(rng, ["sg", "moving_average", "none", "savitzky_golay"])),
           ("td_window_length", number(rng)), ("td_rms_threshold_factor", number(rng)), ("td_taper_fraction_remaining", number(rng)),
           ("td_export_plot_csv", boolean(rng)), ("unknown_key", "17")]
    for kv in opt:
        if rng.random() < 0.25:
            keys.append(kv)
    if rng.random() < 0.1:
        keys = keys[1:]  # no name
    if rng.random() < 0.1 and len(keys) > 1:
        keys.pop(1 if keys[0][0] == "name" else 0)
    if rng.random() < 0.3:
        rng.shuffle(keys)
    first = True
    for k, v in keys:
        sep = ": " if rng.random() < 0.9 else (":" if rng.random() < 0.5 else " : ")
        if first:
            lines.append(f"{pad}- {k}{sep}{v}")
            first = False
        else:
            extra = 2 if rng.random() < 0.93 else int(rng.integers(0, 5))
            lines.append(f"{pad}{' ' * extra}{k}{sep}{v}")
    return lines


def waves(rng, ind):
    pad = " " * ind
    pool = ["regular", "regular", "irregular", "irregular", "still", "no_wave", "still_ci", "none", "Regular", "Irregular", "jonswap", ""]
    typ = str(rng.choice(pool[:8])) if rng.random() < 0.8 else word(rng, pool)
    lines = [f"{pad}type: {typ}"]
    needs = "regular" in typ.lower()
    hk = str(rng.choice(["height", "height", "h", "H", "amplitude", "a", "Hs", "hs"]))
    pk = str(rng.choice(["period", "period", "t", "T", "tp", "Tp", "p"]))
    pv = number(rng) if rng.random() < 0.75 else str(rng.choice(["{ values: [6.0, 7.5, 9] }", "{ linspace: { start: 6, stop: 9, num: 4 } }", "[6, 7]",
                                                                "{values: [5]}", "{ values: [] }", "{ linspace: { start: 6, stop: 9, num: 1 } }"]))
    opt = []
    if rng.random() < (0.93 if needs else 0.5):
        opt.append((hk, number(rng)))
    if rng.random() < (0.93 if needs else 0.5):
        opt.append((pk, pv))
    for kv in (("direction", number(rng)), ("phase", number(rng)), ("seed", number(rng)),
               ("spectrum", word(rng, ["jonswap", "pm", "pierson_moskowitz", "JONSWAP", "bretschneider"])), ("gamma", number(rng))):
        if rng.random() < 0.3:
            opt.append(kv)
    if rng.random() < 0.08:
        opt.append((str(rng.choice(["amplitude", "height", "a", "h"])), number(rng)))  # a second height-like key: consistent or not
    for k, v in opt:
        lines.append(f"{pad}{k}: {v}")
    if rng.random() < 0.2:
        rng.shuffle(lines)
    return lines


def convolution(rng, ind):
    pad = " " * ind
    lines = [f"{pad}mode: {word(rng, ['TaperedDirect', 'Baseline', 'tapered', 'baseline'])}"]
    if rng.random() < 0.6:
        if rng.random() < 0.5:
            lines += [f"{pad}smoothing:", f"{pad}  type: {word(rng, ['moving_average', 'sg', 'none'])}", f"{pad}  window_length: {number(rng)}"]
            if rng.random() < 0.5:
                lines.append(f"{pad}  order: {number(rng)}")
        else:
            lines.append(f"{pad}smoothing: {word(rng, ['moving_average', 'sg'])}")
    if rng.random() < 0.6:
        lines.append(f"{pad}taper:")
        for k in ("start_percent", "end_percent", "final_amplitude", "end_time"):
            if rng.random() < 0.6:
                lines.append(f"{pad}  {k}: {number(rng)}")
    if rng.random() < 0.4:
        lines += [f"{pad}diagnostics:", f"{pad}  export_csv: {boolean(rng)}"]
    return lines


def document(rng):
    unit = int(rng.choice([2, 2, 2, 4, 3]))
    top = []
    if rng.random() < 0.3:
        top.append("# generated")
    if rng.random() < 0.05:
        top.append("model:")
        top.append("  name: x")
    root = "hydrodynamics:" if rng.random() < 0.93 else str(rng.choice(["hydro:", "Hydrodynamics:", "hydrodynamics :", " hydrodynamics:"]))
    top.append(root)
    sections = []
    nb = int(rng.choice([0, 1, 1, 1, 2, 3]))
    blines = [" " * unit + "bodies:"]
    for i in range(nb):
        blines += body(rng, i, 2 * unit)
        if rng.random() < 0.1:
            blines.append("")
    sections.append(blines)
    if rng.random() < 0.9:
        sections.append([" " * unit + "waves:"] + waves(rng, 2 * unit))
    if rng.random() < 0.35:
        sections.append([" " * unit + str(rng.choice(["convolution:", "radiation_convolution:"]))] + convolution(rng, 2 * unit))
    flat = [("radiation_convolution_mode", word(rng, ["TaperedDirect", "Baseline"])), ("td_smoothing", word(rng, ["sg", "moving_average"])),
            ("td_window_length", number(rng)), ("td_export_plot_csv", boolean(rng)), ("td_rirf_end_time", number(rng)),
            ("td_taper_start_percent", number(rng)), ("td_taper_end_percent", number(rng)), ("td_taper_final_amplitude", number(rng))]
    fl = [" " * unit + f"{k}: {v}" for k, v in flat if rng.random() < 0.15]
    if fl:
        sections.append(fl)
    if rng.random() < 0.4:
        rng.shuffle(sections)
    lines = top + [ln for sec in sections for ln in sec]
    # damage
    r = rng.random()
    if r < 0.05 and len(lines) > 3:
        lines.pop(int(rng.integers(1, len(lines))))
    elif r < 0.10 and len(lines) > 3:
        k = int(rng.integers(1, len(lines)))
        lines[k] = " " * int(rng.integers(0, 7)) + lines[k].lstrip()
    elif r < 0.13 and len(lines) > 3:
        k = int(rng.integers(1, len(lines)))
        lines.insert(k, lines[k])
    elif r < 0.15:
        lines.insert(int(rng.integers(1, len(lines) + 1)), str(rng.choice(["???", "  - ", "key without colon", "\t tabbed: 1", ":", "  : 3", "  a: b: c"])))
    elif r < 0.17:
        lines = [ln.replace("  ", "\t", 1) if rng.random() < 0.2 else ln for ln in lines]
    eol = "\n" if rng.random() < 0.95 else "\r\n"
    text = eol.join(lines) + (eol if rng.random() < 0.9 else "")
    return text


devnull = os.open(os.devnull, os.O_WRONLY)
os.dup2(devnull, 2)  # (the reference parser warns on stderr for every document without bodies)
tmp = tempfile.mkdtemp(prefix="hc_fuzz_yaml_")
path = os.path.join(tmp, "case.hydro.yaml")
verdicts = {"ok": 0, "error": 0}
for seed in range(seed0, seed0 + ncases):
    rng = np.random.default_rng(seed)
    text = document(rng)
    with open(path, "wb") as fh:
        fh.write(text.encode())
    try:
        verdicts[T.compare(path)] += 1
    except (AssertionError, Exception) as e:  # noqa: BLE001
        keep = os.path.join(tmp, f"fail_seed_{seed}.hydro.yaml")
        os.replace(path, keep)
        print(f"FAIL seed {seed}: {type(e).__name__}: {str(e)[:600]}\n--- document ({keep}) ---\n{text}\n---")
        ref, ref_err = T.ref_parse(keep)
        got, got_err = T.ours_parse(keep)
        print("reference:", ref if ref is not None else f"ERROR {ref_err}")
        print("ours     :", got if got is not None else f"ERROR {got_err}")
        sys.exit(1)
print(f"fuzz ok: {ncases} documents (seeds {seed0} .. {seed0 + ncases - 1}): accepted alike {verdicts['ok']}, rejected alike with the same message {verdicts['error']}")
