"""Which checkpoint the reference resumes from, and which ones it deletes before saving a new one -- decided by the reference's OWN
statements (/root/reference/train_svd.py:901-925 and :1062-1090), lifted out of `main()` and executed here on scratch directories
with stubs for accelerate / logging.  The decisions are stored in checkpoint_rules.json; tests/test_resume_ema.py holds
svd_xtend_amd.checkpoint.latest_checkpoint / global_step_of / rotate_checkpoints to them.
Usage (this container only): python tests/golden/make_golden_checkpoint_rules.py
"""
import ast
import json
import os
import shutil
import tempfile
from types import SimpleNamespace

REF = "/root/reference/train_svd.py"
# (existing folders, --resume_from_checkpoint, --checkpoints_total_limit, global_step at the save)
CASES = [
    (["checkpoint-500", "checkpoint-1000", "checkpoint-1500", "checkpoint-10000", "logs"], "latest", 3, 10500),
    (["checkpoint-500", "checkpoint-1000"], "latest", 2, 1500),
    (["checkpoint-500", "checkpoint-1000"], "some/where/checkpoint-500", 5, 1500),
    (["checkpoint-9", "checkpoint-10", "checkpoint-100"], "latest", 1, 110),
    (["logs"], "latest", 2, 500),
    (["checkpoint-2", "checkpoint-4", "checkpoint-6"], "latest", None, 8),
]


def lift(tree, lo, hi):
    inside = [n for n in ast.walk(tree) if isinstance(n, ast.stmt) and n.lineno >= lo and n.end_lineno <= hi]
    top = sorted([n for n in inside if not any(m is not n and any(c is n for c in ast.walk(m)) for m in inside)], key=lambda n: n.lineno)
    return compile(ast.Module(body=top, type_ignores=[]), REF, "exec")


def main():
    tree = ast.parse(open(REF).read())
    resume, rotate = lift(tree, 901, 926), lift(tree, 1062, 1091)
    quiet = SimpleNamespace(info=lambda *a, **k: None)
    out = []
    for dirs, resume_arg, limit, step in CASES:
        with tempfile.TemporaryDirectory() as root:
            for d in dirs:
                os.makedirs(os.path.join(root, d))
            loaded = []
            acc = SimpleNamespace(print=lambda *a, **k: None, load_state=loaded.append, save_state=lambda p: os.makedirs(p))
            args = SimpleNamespace(resume_from_checkpoint=resume_arg, output_dir=root, gradient_accumulation_steps=2,
                                   checkpointing_steps=1, checkpoints_total_limit=limit)
            ns = dict(os=os, shutil=shutil, args=args, accelerator=acc, logger=quiet, global_step=0, first_epoch=0,
                      num_update_steps_per_epoch=300)
            exec(resume, ns)
            rec = dict(dirs=dirs, resume_from_checkpoint=resume_arg, checkpoints_total_limit=limit, save_at_step=step,
                       loaded=[os.path.relpath(p, root) for p in loaded], global_step=ns["global_step"], first_epoch=ns["first_epoch"],
                       resume_step=ns.get("resume_step"), resume_arg_after=args.resume_from_checkpoint)
            before = set(os.listdir(root))
            ns["global_step"] = step
            exec(rotate, ns)
            after = set(os.listdir(root))
            rec["removed"] = sorted(before - after, key=lambda d: int(d.split("-")[1]))
            rec["created"] = sorted(after - before)
            out.append(rec)
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "checkpoint_rules.json")
    with open(path, "w") as f:
        json.dump(out, f, indent=1)
    for r in out:
        print(r["resume_from_checkpoint"], r["loaded"], r["global_step"], r["first_epoch"], r["resume_step"], "removed", r["removed"], "created", r["created"])


if __name__ == "__main__":
    main()
