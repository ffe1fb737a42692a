"""The stand-in Chrono headers of tests/cpp/chrono_stub/ are only as good as their agreement with what the reference really uses.
Project Chrono is not installed in the build image, so this is as far as rows a14 / b can be taken: an inventory of every Chrono
type, member function (with the number of arguments it is called with), enumerator and overridden virtual in the reference's
hot-path files (tests/golden/chrono_usage.json, extracted by tests/golden/make_chrono_usage.py) against the stub --

  * everything the hot path uses is declared in the stub, callable with that many arguments / overridable with that many parameters;
  * the stub declares NOTHING ELSE: every PascalCase member is one the reference (hot path or the driver lines the drop-in test
    repeats) calls, or a stub-only helper that says so -- the binding cannot lean on an invented Chrono API;
  * every declaration carries the reference line it stands for.

It pins names and arities, not Chrono 9.0.1's parameter types or semantics (ChLoadCustomMultiple's constructor, ChLoadJacobians,
ChForce::AlignmentFrame, GetNumCoordsVelLevel ...): those stay unverified until a real Chrono build exists."""
import json
import os
import re
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
STUB = os.path.join(ROOT, "tests", "cpp", "chrono_stub")
USAGE = os.path.join(ROOT, "tests", "golden", "chrono_usage.json")
REFERENCE = "/root/reference"

# members of the stub that are not Chrono API: helpers that stand for a Chrono-internal pass, each marked "stub-only" where declared
STUB_ONLY = {"Evaluate": "what ChForce::UpdateTime does with its three modulation functions",
             "StubUpdate": "ChLoadBase::Update -> CreateJacobianMatrices + ComputeJacobian"}
# Eigen::Matrix API of ChMatrixDynamic / ChVectorDynamic (Eigen types in Chrono); lowercase, so outside the PascalCase audit, listed
# for the record: the reference uses setZero / block / rows / size (chrono_usage.json), the binding also operator(), data(), cols()
OWN_TYPES = {"ChLoadAddedMass"}  # the reference's own class


def stub_text():
    parts = []
    for dirpath, _, files in os.walk(STUB):
        for f in sorted(files):
            if f.endswith(".h"):
                parts.append((os.path.relpath(os.path.join(dirpath, f), STUB), open(os.path.join(dirpath, f)).read()))
    return parts


def stub_declarations():
    """{name: [(file, line text, parameter count range)]} for every PascalCase function declared in the stub, and the declared types."""
    decls, types, enumerators = {}, set(), set()
    for rel, text in stub_text():
        for ln in text.splitlines():
            code = ln.split("//")[0]
            for m in re.finditer(r"\b(?:class|struct)\s+(Ch\w+)", code):
                types.add(m.group(1))
            m = re.search(r"enum class (\w+)\s*\{([^}]*)\}", code)
            if m:
                for e in m.group(2).split(","):
                    enumerators.add(f"{m.group(1)}::{e.strip()}")
            # a declaration or inline definition: return type, name, parameter list, then ; { const = override
            for m in re.finditer(r"(?:^|[\s*&>])([A-Z]\w*)\s*\(([^()]*(?:\([^()]*\)[^()]*)*)\)\s*(?:const\s*)?(?:override\s*)?(?:=\s*0\s*)?[;{:]", code):
                name, params = m.group(1), m.group(2)
                before = code[:m.start(1)]
                if re.search(r"(->|\.)\s*$", before) or re.search(r"\breturn\b[^;]*$", before):
                    continue  # a call, not a declaration
                has_type_in_front = re.search(r"[\w>&*]\s+$", before) is not None and not re.search(r"\b(?:return|new|else)\s+$", before)
                is_constructor = before.strip() in ("", "explicit") and re.fullmatch(r"Ch\w+|Block", name) is not None
                if not (has_type_in_front or is_constructor):
                    continue  # a statement that calls the function
                plist = [p for p in re.split(r",(?![^<]*>)", params) if p.strip()]
                n_max = len(plist)
                n_min = sum(1 for p in plist if "=" not in p)
                decls.setdefault(name, []).append((rel, ln.strip(), n_min, n_max))
    return decls, types, enumerators


def test_committed_inventory_is_what_the_reference_holds():
    """Where the reference tree is present (the build container), the committed inventory is re-extracted and must be unchanged."""
    if not os.path.isdir(REFERENCE):
        pytest.skip("no reference tree here: the committed inventory stands")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "golden", "make_chrono_usage.py"), "--print"], capture_output=True, text=True, check=True)
    assert json.loads(r.stdout) == json.load(open(USAGE)), "tests/golden/chrono_usage.json is stale: run tests/golden/make_chrono_usage.py"


def test_stub_declares_everything_the_hot_path_uses_with_its_arity():
    use = json.load(open(USAGE))["hot_path"]
    decls, types, enumerators = stub_declarations()
    all_text = "\n".join(t for _, t in stub_text())
    for t, where in use["types"].items():
        if t in OWN_TYPES:
            continue
        if "::" in t:  # a field of a Chrono struct (m_jacobians->M)
            owner, field = t.split("::")
            m = re.search(r"struct\s+" + owner + r"\s*\{([^}]*)\}", all_text, re.S)
            assert m and re.search(r"\b" + field + r"\b", m.group(1)), f"{t} ({where[0]}) is not a member of the stub's {owner}"
            continue
        assert t in types, f"the reference uses chrono::{t} ({where[0]}); the stub does not declare it"
    for name, by_arity in use["member_calls"].items():
        if name[0].islower():
            assert re.search(r"\b" + name + r"\s*\(", all_text), f"Eigen-style member {name}() ({list(by_arity.values())[0][0]}) is missing from the stub"
            continue
        assert name in decls, f"the reference calls {name}() ({list(by_arity.values())[0][0]}); the stub does not declare it"
        for arity, where in by_arity.items():
            ok = any(lo <= int(arity) <= hi for _, _, lo, hi in decls[name])
            assert ok, f"{name} is called with {arity} argument(s) at {where[0]}; the stub declares {[(d[2], d[3]) for d in decls[name]]}"
    for e, where in use["enumerators"].items():
        assert e.split("::", 1)[1] in enumerators, f"{e} ({where[0]}) is not an enumerator of the stub"
    for name, by_arity in use["overrides"].items():
        virt = [d for d in decls.get(name, []) if "virtual" in d[1]]
        assert virt, f"the reference overrides {name} ({list(by_arity.values())[0][0]}); the stub declares no such virtual"
        for arity, where in by_arity.items():
            assert any(d[3] == int(arity) for d in virt), f"{name} is overridden with {arity} parameter(s) at {where[0]}; the stub's virtual takes {[d[3] for d in virt]}"


def test_stub_declares_nothing_the_reference_does_not_use_and_cites_every_declaration():
    usage = json.load(open(USAGE))
    allowed = set(STUB_ONLY)
    for key in ("hot_path", "drivers"):
        allowed |= {n for n in usage[key]["member_calls"] if n[0].isupper()} | set(usage[key]["overrides"])
        allowed |= {t for t in usage[key]["types"] if "::" not in t}  # constructors
    allowed |= {"ChQuaterniond", "ChLoadBase", "ChLoadJacobians", "Block"}  # types the used calls return / derive from (cited where declared)
    decls, types, _ = stub_declarations()
    for name, where in decls.items():
        assert name in allowed, f"the stub declares {name} ({where[0][0]}: {where[0][1]}), which the reference never calls"
        for rel, line, _, _ in where:
            if name in STUB_ONLY or name in ("Block",):
                continue
            cited = re.search(r"ref: [\w/.]+\.(?:cpp|h):\d+", line)
            assert cited or rel.endswith("ChStubTypes.h") and name in ("ChVector3d", "ChVectorDynamic", "ChMatrixDynamic"), \
                f"{rel}: `{line}` does not cite the reference line it stands for"
    for rel, text in stub_text():
        for helper, why in STUB_ONLY.items():
            if re.search(r"\b" + helper + r"\s*\(", text) and re.search(r"\b(?:void|ChVector3d)\s+" + helper + r"\s*\(", text):
                k = text.index(helper)
                assert "stub-only" in text[max(0, k - 300):k], f"{rel}: {helper} must be marked stub-only ({why})"
    # every cited line exists in the inventory's files (a citation is file:line of a file the extractor read, or a header range)
    files = {f.split(":")[0] for key in ("hot_path", "drivers") for f in usage[key]["files"]}
    for rel, text in stub_text():
        for m in re.finditer(r"ref: ([\w/.]+\.(?:cpp|h)):\d+", text):
            assert m.group(1) in files, f"{rel} cites {m.group(1)}, which is not among the audited reference files"
