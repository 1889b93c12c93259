#!/usr/bin/env python3
"""API-surface fixture: what the reference's CALLERS touch on the objects this repository replaces.

TEST INFRASTRUCTURE ONLY.  Runs in the build container (reads /root/reference as text through `ast` -- nothing
is imported or executed) and writes tests/golden/api_surface.json, which tests/test_api_surface.py checks
against this repository's classes on CPU.  The fixture is data: names, argument counts, keyword names and
signatures -- no source text.

Two parts:
  uses        for every caller file (scripts/train.py, src/evaluation/evaluator.py, and the towers' use of the
              id container), every attribute / method the file touches on a variable with a known role
              (task, evaluator, loader, schema, kjt, result dict, prediction dict), with positional-argument
              counts and keyword names of calls, the string keys subscripted on dict-like roles, and the
              `from ... import ...` lines the driver needs to resolve;
  signatures  parameter names + default expressions of the reference's public definitions on the path
              (TwoTowerEvaluator, TwoTowerTrainTask, TwoTowerModel, the factories, the schema builder, the
              dataloader factory, FeaturePreprocessor, FeatureProjector).

Usage:  python oracle/gen_api_surface.py
"""
from __future__ import annotations

import ast
import json
import os
from pathlib import Path

ROOT = Path(__file__).resolve().parents[1]
REF = Path(os.environ.get("TT_REFERENCE_ROOT", "/root/reference"))
OUT = ROOT / "tests" / "golden" / "api_surface.json"

# caller file -> {variable name: role}
CALLERS = {
    "scripts/train.py": {"train_task": "task", "model": "task", "evaluator": "evaluator", "train_loader": "loader",
                         "test_loader": "loader", "schema": "schema", "result": "result", "test_metrics": "metrics",
                         "sample_metrics": "metrics"},
    "src/evaluation/evaluator.py": {"model": "task", "self": "evaluator", "dataloader": "loader", "result": "result",
                                    "predictions": "predictions", "metrics": "metrics"},
    "src/towers/cat_embed.py": {"kjt": "kjt"},
    "src/towers/tower/base_tower.py": {"kjt": "kjt"},
}
# definitions whose signatures the drop-in must accept: file -> top-level names (classes: every public method)
DEFINITIONS = {
    "src/evaluation/evaluator.py": ["TwoTowerEvaluator"],
    "src/towers/two_tower_train_task.py": ["TwoTowerTrainTask", "create_two_tower_train_task"],
    "src/towers/two_tower_model.py": ["TwoTowerModel", "create_two_tower_model"],
    "src/torchrec_preprocess/schema.py": ["build_torchrec_schema_from_meta"],
    "src/towers/pairs/unified_bid_data_loader.py": ["create_unified_bid_dataloaders"],
    "src/torchrec_preprocess/feature_preprocessor.py": ["FeaturePreprocessor"],
    "src/torchrec_preprocess/feature_projector.py": ["FeatureProjector"],
}


def _base_name(node):
    """Name at the root of an attribute / subscript chain, and the chain of attribute names below it."""
    chain = []
    while isinstance(node, (ast.Attribute, ast.Subscript, ast.Call)):
        if isinstance(node, ast.Attribute):
            chain.append(node.attr)
            node = node.value
        elif isinstance(node, ast.Subscript):
            node = node.value
        else:
            node = node.func
    return (node.id if isinstance(node, ast.Name) else None), chain[::-1]


def scan_uses(path: str, roles):
    tree = ast.parse((REF / path).read_text(encoding="utf-8"))
    uses = {}

    def rec(role):
        return uses.setdefault(role, {"attributes": set(), "calls": {}, "keys": set(), "called_directly": []})

    for node in ast.walk(tree):
        if isinstance(node, ast.Attribute) and isinstance(node.value, ast.Name) and node.value.id in roles:
            rec(roles[node.value.id])["attributes"].add(node.attr)
        if isinstance(node, ast.Call):
            f = node.func
            n_pos, kws = len(node.args), sorted(k.arg for k in node.keywords if k.arg)
            if isinstance(f, ast.Attribute) and isinstance(f.value, ast.Name) and f.value.id in roles:
                c = rec(roles[f.value.id])["calls"].setdefault(f.attr, [])
                if [n_pos, kws] not in c:
                    c.append([n_pos, kws])
            elif isinstance(f, ast.Name) and f.id in roles:                      # model(batch, return_metrics=True)
                c = rec(roles[f.id])["called_directly"]
                if [n_pos, kws] not in c:
                    c.append([n_pos, kws])
            elif isinstance(f, ast.Name) and f.id in ("len", "iter") and node.args and isinstance(node.args[0], ast.Name) \
                    and node.args[0].id in roles:
                rec(roles[node.args[0].id])["attributes"].add(f"__{f.id}__")
            # batch["notice"]["kjt"].to(device) / tower_input["kjt"].to(...): the id container reached through a batch dict
            if isinstance(f, ast.Attribute) and isinstance(f.value, ast.Subscript) and isinstance(f.value.slice, ast.Constant) \
                    and f.value.slice.value == "kjt":
                r = rec("kjt")
                r["attributes"].add(f.attr)
                c = r["calls"].setdefault(f.attr, [])
                if [n_pos, kws] not in c:
                    c.append([n_pos, kws])
            # nested: schema.notice.categorical
            if isinstance(f, ast.Attribute):
                base, chain = _base_name(f)
                if base in roles and len(chain) > 1:
                    rec(roles[base])["attributes"].add(".".join(chain[:-1]))
        if isinstance(node, ast.Attribute):
            base, chain = _base_name(node)
            if base in roles and len(chain) > 1:
                rec(roles[base])["attributes"].add(".".join(chain))
        if isinstance(node, ast.Subscript) and isinstance(node.value, ast.Name) and node.value.id in roles:
            s = node.slice
            if isinstance(s, ast.Constant) and isinstance(s.value, str):
                rec(roles[node.value.id])["keys"].add(s.value)
        if isinstance(node, ast.Call) and isinstance(node.func, ast.Attribute) and node.func.attr == "get" \
                and isinstance(node.func.value, ast.Name) and node.func.value.id in roles and node.args \
                and isinstance(node.args[0], ast.Constant) and isinstance(node.args[0].value, str):
            rec(roles[node.func.value.id])["keys"].add(node.args[0].value)
        if isinstance(node, ast.For) and isinstance(node.iter, ast.Name) and node.iter.id in roles:
            rec(roles[node.iter.id])["attributes"].add("__iter__")
    imports = []
    for node in tree.body:
        if isinstance(node, ast.ImportFrom) and node.module and node.module.split(".")[0] == "src":
            imports.append({"module": node.module, "names": [a.name for a in node.names]})
    out = {role: {"attributes": sorted(u["attributes"]), "calls": {k: u["calls"][k] for k in sorted(u["calls"])},
                  "keys": sorted(u["keys"]), "called_directly": u["called_directly"]} for role, u in sorted(uses.items())}
    return out, imports


def _sig(fn: ast.FunctionDef):
    a = fn.args
    pos = [x.arg for x in a.posonlyargs + a.args]
    defaults = [None] * (len(pos) - len(a.defaults)) + [ast.unparse(d) for d in a.defaults]
    return {"params": [{"name": n, "default": d} for n, d in zip(pos, defaults)],
            "kwonly": [{"name": x.arg, "default": (ast.unparse(d) if d is not None else None)}
                       for x, d in zip(a.kwonlyargs, a.kw_defaults)],
            "varargs": a.vararg is not None, "varkw": a.kwarg is not None}


def scan_definitions(path: str, names):
    tree = ast.parse((REF / path).read_text(encoding="utf-8"))
    out = {}
    for node in tree.body:
        if isinstance(node, ast.FunctionDef) and node.name in names:
            out[node.name] = _sig(node)                     # a later def of the same name replaces an earlier one, as at import
        if isinstance(node, ast.ClassDef) and node.name in names:
            methods = {}
            for m in node.body:
                if isinstance(m, ast.FunctionDef) and (not m.name.startswith("_") or m.name == "__init__"):
                    methods[m.name] = _sig(m)
            out[node.name] = {"bases": [ast.unparse(b) for b in node.bases], "methods": methods}
    return out


def scan_harness(path: str = "scripts/train.py"):
    """The driver's own output contracts ("the results-CSV columns (:37-57)", "checkpoint dict keys (:506-511)", "config
    keys (:84-134)"): keys of the dict literals named result_row / hyperparams / final_metrics / config / checkpoint, in source
    order, and for result_row which dict and key each column is read from."""
    tree = ast.parse((REF / path).read_text(encoding="utf-8"))
    want = {"result_row": "results_csv", "hyperparams": "hyperparams_keys", "final_metrics": "final_metrics_keys", "config": "config_keys",
            "checkpoint": "checkpoint_keys"}
    out = {}
    for node in ast.walk(tree):
        if isinstance(node, ast.Assign) and len(node.targets) == 1 and isinstance(node.targets[0], ast.Name) \
                and node.targets[0].id in want and isinstance(node.value, ast.Dict):
            name = node.targets[0].id
            keys = [k.value for k in node.value.keys if isinstance(k, ast.Constant)]
            if name == "result_row":
                cols = []
                for k, v in zip(node.value.keys, node.value.values):
                    src = None
                    for sub in ast.walk(v):                  # hyperparams.get("batch_size", "N/A") -> ["hyperparams", "batch_size"]
                        if isinstance(sub, ast.Call) and isinstance(sub.func, ast.Attribute) and sub.func.attr == "get" \
                                and isinstance(sub.func.value, ast.Name) and sub.args and isinstance(sub.args[0], ast.Constant):
                            src = [sub.func.value.id, sub.args[0].value]
                    cols.append({"column": k.value, "read_from": src})
                out["results_csv"] = {"columns": cols}
            else:
                out.setdefault(want[name], keys)
        if isinstance(node, ast.FunctionDef) and node.name == "save_training_results":
            a = node.args
            out["results_csv_default_file"] = ast.literal_eval(a.defaults[-1]) if a.defaults else None
    return out


def main():
    surface = {"generated_by": "oracle/gen_api_surface.py (ast walk of the reference; nothing imported)", "uses": {}, "imports": {},
               "signatures": {}, "harness": {"scripts/train.py": scan_harness()}}
    for path, roles in CALLERS.items():
        uses, imports = scan_uses(path, roles)
        surface["uses"][path] = uses
        if imports:
            surface["imports"][path] = imports
    for path, names in DEFINITIONS.items():
        surface["signatures"][path] = scan_definitions(path, names)
    OUT.write_text(json.dumps(surface, indent=1, ensure_ascii=False, sort_keys=True) + "\n", encoding="utf-8")
    n_use = sum(len(u["attributes"]) + len(u["calls"]) for f in surface["uses"].values() for u in f.values())
    print(f"wrote {OUT} ({OUT.stat().st_size} bytes): {n_use} uses, "
          f"{sum(len(v) for v in surface['signatures'].values())} definitions")


if __name__ == "__main__":
    main()
