"""evidence_merge.py <tag> <partial tag> -- fold a partial re-run of the configuration table into the round's committed one:
`EVIDENCE_ONLY=cfgs EVIDENCE_CFGS="cfg ..." bash tests/tools/evidence_round6.sh <partial tag>` on the GPU box leaves
gpurun_out/<partial tag>_cfgs_bench.jsonl and _cfgs_kernel_stats.txt with the named configurations only; their rows and blocks replace the
same configurations' in profiles/<tag>_cfgs_bench.jsonl / _cfgs_kernel_stats.txt (new configurations are appended), and
profiles/<tag>_gainmap_compute.txt is cut from the merged statistics again.  Rows of one file then come from two boxes: the merged file says
which configurations were re-run (a trailing comment line in the statistics file)."""
import json
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent.parent


def blocks_of(text):
    head, *rest = text.split("\n== ")
    return head, {b.split("\n", 1)[0].strip(): b for b in rest}


def main():
    tag, part = sys.argv[1], sys.argv[2]
    prof, out = ROOT / "profiles", ROOT / "gpurun_out"
    new_rows = [l for l in (out / f"{part}_cfgs_bench.jsonl").read_text().splitlines() if l.startswith("{")]
    redone = []
    for l in new_rows:
        c = json.loads(l)["config"]
        if c not in redone:
            redone.append(c)
    old_rows = [l for l in (prof / f"{tag}_cfgs_bench.jsonl").read_text().splitlines() if l.startswith("{")]
    merged, placed = [], set()
    for l in old_rows:
        c = json.loads(l)["config"]
        if c in redone:
            if c not in placed:
                merged += [n for n in new_rows if json.loads(n)["config"] == c]
                placed.add(c)
        else:
            merged.append(l)
    for c in redone:
        if c not in placed:
            merged += [n for n in new_rows if json.loads(n)["config"] == c]
    (prof / f"{tag}_cfgs_bench.jsonl").write_text("\n".join(merged) + "\n")
    head, old_blocks = blocks_of((prof / f"{tag}_cfgs_kernel_stats.txt").read_text())
    _, new_blocks = blocks_of((out / f"{part}_cfgs_kernel_stats.txt").read_text())
    for c, b in new_blocks.items():
        old_blocks[c] = b.rstrip("\n") + "\n"
    note = f"# re-run later in the round on another box (tests/tools/evidence_merge.py {tag} {part}): " + " ".join(redone)
    body = head.rstrip("\n") + "\n" + "".join("\n== " + b.rstrip("\n") + "\n" for b in old_blocks.values())
    body = "\n".join(l for l in body.splitlines() if not l.startswith("# re-run later")) + "\n" + note + "\n"
    (prof / f"{tag}_cfgs_kernel_stats.txt").write_text(body)
    keep, lines = False, []
    for line in body.splitlines(keepends=True):
        if line.startswith("== "):
            keep = line.split()[1] in ("gmcompute4k", "gmcompute4k_dev", "gainmap4k", "gainmap4k_half")
        if keep or line.startswith("rocprofv3") or line.startswith("# re-run"):
            lines.append(line)
    (prof / f"{tag}_gainmap_compute.txt").write_text("".join(lines))
    print(f"merged {len(redone)} configuration(s): {' '.join(redone)}")


if __name__ == "__main__":
    main()
