"""Inventory of the Project Chrono API the reference's hot-path files use -- types, member functions with the number of arguments
they are called with, enumerators, and the virtual functions its classes override -- with the file:line of every use.  Written to
tests/golden/chrono_usage.json (data about the reference, no source text); tests/test_chrono_stub_audit.py holds the stand-in
headers of tests/cpp/chrono_stub/ against it: everything listed must be declared there with that arity, and nothing else may be.

Run in the build container (needs /root/reference):  python tests/golden/make_chrono_usage.py
"""
import json
import os
import re
import sys

REF = os.environ.get("HYDROCHRONO_REFERENCE", "/root/reference")
# the files behind SURVEY 8a / 8b (force path, added-mass load, YAML wiring) ...
HOT = ["src/hydro_forces.cpp", "src/chloadaddedmass.cpp", "include/hydroc/hydro_forces.h", "include/hydroc/chloadaddedmass.h",
       "src/setup_hydro_from_yaml.cpp", "src/setup_hydro_from_yaml.h"]
# ... and the driver lines tests/cpp/chrono_dropin_test.cpp repeats (file, first line, last line): the Chrono calls around the hydro
# lines of the sphere decay / regular-wave drivers and of the YAML runner are what that test's main() needs from the stand-in
DRIVERS = [("tests/regression/sphere/demo_sphere_decay.cpp", 50, 125), ("demos/sphere/demo_sphere_reg_waves.cpp", 55, 150),
           ("src/hydrochrono_runner/run_hydrochrono_from_yaml.cpp", 430, 460), ("tests/chloadaddedmass_t01.cpp", 30, 70)]
# the reference's own headers: PascalCase members declared there are not Chrono's
OWN = ["include/hydroc/hydro_forces.h", "include/hydroc/chloadaddedmass.h", "include/hydroc/wave_types.h", "include/hydroc/h5fileinfo.h",
       "include/hydroc/helper.h", "src/hydro_types.h", "src/hydro_yaml_parser.h", "src/setup_hydro_from_yaml.h"]


def strip_comments(text):
    text = re.sub(r"/\*.*?\*/", lambda m: "\n" * m.group(0).count("\n"), text, flags=re.S)
    text = re.sub(r"//[^\n]*", "", text)
    return re.sub(r"^[ \t]*#[ \t]*include[^\n]*", "", text, flags=re.M)  # header names are not uses


def call_arity(text, open_idx):
    """Number of top-level arguments of the call whose '(' is at open_idx."""
    depth, i, args, any_tok = 0, open_idx, 0, False
    while i < len(text):
        ch = text[i]
        if ch in "([{<" and not (ch == "<" and not re.match(r"[\w:]", text[i - 1] if i else " ")):
            depth += 1
        elif ch in ")]}>" and not (ch == ">" and text[i - 1] in "-="):
            depth -= 1
            if depth == 0:
                return args + (1 if any_tok else 0)
        elif ch == "," and depth == 1:
            args += 1
        elif depth >= 1 and not ch.isspace():
            any_tok = True
        i += 1
    return None


def own_members():
    names = set()
    for f in OWN:
        p = os.path.join(REF, f)
        if not os.path.exists(p):
            continue
        t = strip_comments(open(p).read())
        names |= set(re.findall(r"\b([A-Z]\w+)\s*\([^;{]*\)\s*(?:const\s*)?(?:override\s*)?(?:noexcept\s*)?[;{=]", t))
    return names


def scan(files, own):
    types, methods, enums, overrides = {}, {}, {}, {}
    for f in files:
        lo = hi = None
        if isinstance(f, (tuple, list)):
            f, lo, hi = f
        p = os.path.join(REF, f)
        text = strip_comments(open(p).read())
        if lo is not None:  # keep the line numbering: the lines outside the range become empty
            lines = text.split("\n")
            text = "\n".join(ln if lo <= k + 1 <= hi else "" for k, ln in enumerate(lines))
        line_of = lambda idx: text.count("\n", 0, idx) + 1  # noqa: E731
        for m in re.finditer(r"\b(?:chrono::)?(Ch[A-Z]\w*)\b", text):
            types.setdefault(m.group(1), []).append(f"{f}:{line_of(m.start())}")
        for m in re.finditer(r"\b(Ch[A-Z]\w*(?:::[A-Za-z_]\w*){1,2})\b(?!\s*\()", text):
            parts = m.group(1).split("::")
            if parts[-1][0].isupper() and parts[-1].isupper() or "_" in parts[-1] and parts[-1].upper() == parts[-1]:
                enums.setdefault(m.group(1), []).append(f"{f}:{line_of(m.start())}")
        for m in re.finditer(r"(?:->|\.)\s*([A-Z]\w+)\s*\(", text):
            name = m.group(1)
            if name in own:
                continue
            ar = call_arity(text, m.end() - 1)
            methods.setdefault(name, {}).setdefault(str(ar), []).append(f"{f}:{line_of(m.start())}")
        if f.endswith("src/hydro_forces.cpp"):  # component accessors of ChVector3d
            for m in re.finditer(r"\.\s*([xyz])\s*\(\s*\)", text):
                methods.setdefault(m.group(1), {}).setdefault("0", []).append(f"{f}:{line_of(m.start())}")
        if f.endswith("chloadaddedmass.cpp"):  # Eigen-style members of ChMatrixDynamic / ChVectorDynamic, fields of ChLoadJacobians
            for m in re.finditer(r"(?:->|\.)\s*([a-z]\w*)\s*\(", text):
                ar = call_arity(text, m.end() - 1)
                methods.setdefault(m.group(1), {}).setdefault(str(ar), []).append(f"{f}:{line_of(m.start())}")
            for m in re.finditer(r"m_jacobians\s*->\s*(\w+)", text):
                types.setdefault("ChLoadJacobians::" + m.group(1), []).append(f"{f}:{line_of(m.start())}")
        for m in re.finditer(r"\b([A-Z]\w+)\s*\(([^;{}]*)\)\s*(?:const\s*)?override", text):
            params = [x for x in m.group(2).split(",") if x.strip()]
            overrides.setdefault(m.group(1), {}).setdefault(str(len(params)), []).append(f"{f}:{line_of(m.start())}")
    return types, methods, enums, overrides


def main():
    own = own_members()
    out = {"reference": "Project-SEA-Stack/HydroChrono @ 2025-10-31", "generated_by": "tests/golden/make_chrono_usage.py"}
    for key, files in (("hot_path", HOT), ("drivers", DRIVERS)):
        types, methods, enums, overrides = scan(files, own)
        out[key] = {"files": [f if isinstance(f, str) else f"{f[0]}:{f[1]}-{f[2]}" for f in files], "types": {k: sorted(set(v))[:6] for k, v in sorted(types.items())},
                    "member_calls": {k: {a: sorted(set(l))[:6] for a, l in sorted(v.items())} for k, v in sorted(methods.items())},
                    "enumerators": {k: sorted(set(v))[:6] for k, v in sorted(enums.items())},
                    "overrides": {k: {a: sorted(set(l))[:6] for a, l in sorted(v.items())} for k, v in sorted(overrides.items())}}
    dst = os.path.join(os.path.dirname(os.path.abspath(__file__)), "chrono_usage.json")
    if "--print" in sys.argv:
        print(json.dumps(out, indent=1))
    else:
        json.dump(out, open(dst, "w"), indent=1, sort_keys=True)
        print(dst)


if __name__ == "__main__":
    main()
