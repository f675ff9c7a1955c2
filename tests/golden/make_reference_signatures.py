"""Writes tests/golden/reference_signatures.json: the public call surface of the reference's hot-path modules.

The reference cannot be imported here (jax / numpyro are absent), so its files are PARSED with ``ast`` -
no reference code runs and none is stored: the fixture holds interface data only (module, class, function name,
parameter names in order, kind, and the source text of each default literal).  ``tests/test_signatures_cpu.py``
compares ``inspect.signature`` of every counterpart in ``bobe_amd`` with it.

Run in the container that holds /root/reference:   python tests/golden/make_reference_signatures.py
"""
import ast
import json
import os
import sys

REF = os.environ.get("BOBE_REFERENCE", "/root/reference")
MODULES = ["gp", "bo", "acquisition", "clf_gp", "clf", "samplers", "optim", "pool", "likelihood", "utils.core",
           "utils.seed", "utils.log"]
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "reference_signatures.json")


def params_of(fn: ast.FunctionDef):
    a = fn.args
    out = []
    pos = list(a.posonlyargs) + list(a.args)
    defaults = [None] * (len(pos) - len(a.defaults)) + list(a.defaults)
    for p, d in zip(pos, defaults):
        out.append({"name": p.arg, "kind": "positional_or_keyword",
                    "default": None if d is None else ast.unparse(d)})
    if a.vararg is not None:
        out.append({"name": a.vararg.arg, "kind": "var_positional", "default": None})
    for p, d in zip(a.kwonlyargs, a.kw_defaults):
        out.append({"name": p.arg, "kind": "keyword_only", "default": None if d is None else ast.unparse(d)})
    if a.kwarg is not None:
        out.append({"name": a.kwarg.arg, "kind": "var_keyword", "default": None})
    return out


def decorators(fn):
    return [ast.unparse(d) for d in fn.decorator_list]


def main():
    sigs = {}
    for mod in MODULES:
        path = os.path.join(REF, "BOBE", *mod.split(".")) + ".py"
        tree = ast.parse(open(path).read(), filename=path)
        entry = {"functions": {}, "classes": {}}
        for node in tree.body:
            if isinstance(node, ast.FunctionDef) and not node.name.startswith("_"):
                entry["functions"][node.name] = {"line": node.lineno, "params": params_of(node)}
            elif isinstance(node, ast.ClassDef):
                methods = {}
                for m in node.body:
                    if isinstance(m, ast.FunctionDef) and (not m.name.startswith("_") or m.name == "__init__"):
                        methods[m.name] = {"line": m.lineno, "params": params_of(m), "decorators": decorators(m)}
                entry["classes"][node.name] = {"line": node.lineno, "bases": [ast.unparse(b) for b in node.bases],
                                               "methods": methods}
        sigs[mod] = entry
    with open(OUT, "w") as fh:
        json.dump({"source": "ast.parse of BOBE/{%s}.py (Ameek94/BOBE as held in /root/reference)" % ",".join(MODULES),
                   "modules": sigs}, fh, indent=1, sort_keys=True)
    n = sum(len(e["functions"]) + sum(len(c["methods"]) for c in e["classes"].values()) for e in sigs.values())
    print("wrote", OUT, n, "callables")


if __name__ == "__main__":
    sys.exit(main())
