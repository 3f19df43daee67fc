#!/usr/bin/env python3
"""Record the import surface the reference's entry scripts use: every `from svg... import name` line of the top-level
scripts of /root/reference, and for every imported *function* its parameter list (names, order, defaults as source text)
taken from the reference module by AST — nothing of the reference is imported or executed.

    python tests/golden/make_golden_api.py      -> tests/golden/api_surface.json
"""
import ast
import json
from pathlib import Path

REF = Path("/root/reference")
SCRIPTS = ["cog_inference.py", "cosmos_t2v_inference.py", "hyvideo_i2v_inference.py", "hyvideo_t2v_inference.py",
           "wan_i2v_inference.py", "wan_t2v_inference.py"]   # orig_hyvideo_inference.py drives the deprecated *_orig fork: out of scope


def module_file(mod: str) -> Path:
    p = REF / Path(*mod.split("."))
    return p.with_suffix(".py") if p.with_suffix(".py").exists() else p / "__init__.py"


def find_def(mod: str, name: str, depth: int = 0):
    """(kind, params) of `name` in reference module `mod`, following one level of re-export."""
    tree = ast.parse(module_file(mod).read_text())
    for node in tree.body:
        if isinstance(node, ast.FunctionDef) and node.name == name:
            a = node.args
            pos = a.posonlyargs + a.args
            defaults = [None] * (len(pos) - len(a.defaults)) + [ast.unparse(d) for d in a.defaults]
            params = [[p.arg, d] for p, d in zip(pos, defaults)]
            params += [[p.arg, ast.unparse(d) if d is not None else None] for p, d in zip(a.kwonlyargs, a.kw_defaults)]
            return "function", params
        if isinstance(node, ast.ClassDef) and node.name == name:
            return "class", None
        if isinstance(node, ast.ImportFrom) and depth < 2 and any(al.name == name for al in node.names):
            base = mod.split(".")
            tgt = ".".join(base[: len(base) - node.level] + ([node.module] if node.module else [])) if node.level else node.module
            if tgt.startswith("svg"):
                return find_def(tgt, name, depth + 1)
        if isinstance(node, ast.Assign) and any(isinstance(t, ast.Name) and t.id == name for t in node.targets):
            return "value", None
    return "missing", None


out = []
for s in SCRIPTS:
    for node in ast.walk(ast.parse((REF / s).read_text())):
        if isinstance(node, ast.ImportFrom) and node.module and node.module.startswith("svg"):
            for al in node.names:
                kind, params = find_def(node.module, al.name)
                out.append({"script": s, "line": node.lineno, "module": node.module, "name": al.name, "kind": kind, "params": params})
Path(__file__).with_name("api_surface.json").write_text(json.dumps(out, indent=1) + "\n")
print(len(out), "imports recorded;", sum(o["kind"] == "missing" for o in out), "unresolved")


# ---- the whole public surface of the non-deprecated package: module -> top-level functions / classes (names only) ----
# (`*_orig` forks, the CUDA extension sources, tests and third-party trees are not part of the path)
surface = {}
for f in sorted((REF / "svg").rglob("*.py")):
    rel = f.relative_to(REF / "svg")
    parts = rel.parts
    if any(p.endswith("_orig") for p in parts) or "3rdparty" in parts or "test" in parts or "csrc" in parts or f.name == "__init__.py":
        continue
    names = [n.name for n in ast.parse(f.read_text()).body
             if isinstance(n, (ast.FunctionDef, ast.ClassDef)) and not n.name.startswith(("_", "test", "benchmark"))]
    surface["svg." + ".".join(rel.with_suffix("").parts)] = names
Path(__file__).with_name("api_public_names.json").write_text(json.dumps(surface, indent=1) + "\n")
print(len(surface), "modules,", sum(len(v) for v in surface.values()), "public names recorded")
