#!/usr/bin/env python3
"""Aggregates rocprofv3's PC-sampling tables (csv) on the GPU box: samples per (kernel, instruction offset) and per every categorical
column the sampler adds (issue / stall reasons of stochastic sampling), for the kernels that hold most samples. usage: RAW_DIR OUT_DIR"""
import collections
import csv
import json
import sys
from pathlib import Path

raw, out = Path(sys.argv[1]), Path(sys.argv[2])
out.mkdir(parents=True, exist_ok=True)
tables = sorted(raw.glob("**/*pc_sampling*.csv"))
print("tables:", [str(t) for t in tables])
kernel_names = {}
for trace in raw.glob("**/*kernel_trace.csv"):
    with open(trace) as handle:
        for row in csv.DictReader(handle):
            kernel_names[row.get("Correlation_Id") or row.get("Dispatch_Id")] = row.get("Kernel_Name", "")
            kernel_names["d" + str(row.get("Dispatch_Id"))] = row.get("Kernel_Name", "")
for table in tables:
    with open(table) as handle:
        head = [next(handle, "") for _ in range(12)]
    (out/f"{table.stem}.head.txt").write_text("".join(head))
    with open(table) as handle:
        reader = csv.DictReader(handle)
        columns = reader.fieldnames or []
        print(table.name, "columns:", columns)
        offset_col = next((c for c in columns if "offset" in c.lower()), None)
        inst_col = next((c for c in columns if c.lower() in ("instruction", "inst", "instruction_comment")), None)
        id_cols = [c for c in columns if c.lower() in ("dispatch_id", "correlation_id")]
        categorical = [c for c in columns if c not in (offset_col, inst_col) and not any(k in c.lower() for k in ("timestamp", "exec_mask", "dispatch", "correlation", "id", "wave", "chiplet", "hw_"))]
        per_offset = collections.defaultdict(collections.Counter)
        per_category = {c: collections.defaultdict(collections.Counter) for c in categorical}
        text = {}
        total = 0
        for row in reader:
            total += 1
            kernel = ""
            for c in id_cols:
                kernel = kernel_names.get(row[c]) or kernel_names.get("d" + row[c]) or kernel
            key = row.get(offset_col, "?") if offset_col else "?"
            per_offset[kernel][key] += 1
            if inst_col:
                text[(kernel, key)] = row[inst_col]
            for c in categorical:
                per_category[c][kernel][row[c]] += 1
        print("samples:", total)
        ranking = sorted(per_offset, key=lambda k: -sum(per_offset[k].values()))[:3]
        result = {}
        for kernel in ranking:
            result[kernel] = {"samples": sum(per_offset[kernel].values()),
                              "by_offset": {k: [v, text.get((kernel, k), "")] for k, v in sorted(per_offset[kernel].items(), key=lambda kv: (len(kv[0]), kv[0]))},
                              "by_category": {c: dict(per_category[c][kernel].most_common(40)) for c in categorical}}
            print(kernel[:100], result[kernel]["samples"], {c: dict(per_category[c][kernel].most_common(6)) for c in categorical})
        (out/f"{table.stem}.aggregated.json").write_text(json.dumps(result))
