#!/bin/bash
# Host topology of a GPU box: NUMA nodes, which node the GPU hangs off, what this process may run on (tools/README)
echo "== nproc: $(nproc) (online: $(cat /sys/devices/system/cpu/online))"
lscpu | grep -E 'Model name|Socket|NUMA|Thread|Core'
echo "== allowed cpus: $(grep Cpus_allowed_list /proc/self/status)"
echo "== allowed mems: $(grep Mems_allowed_list /proc/self/status)"
for n in /sys/devices/system/node/node*; do echo "$(basename $n): cpus $(cat $n/cpulist) mem $(grep MemTotal $n/meminfo | awk '{print $4/1048576 " GiB"}') free $(grep MemFree $n/meminfo | awk '{print $4/1048576 " GiB"}')"; done
echo "== distances"; cat /sys/devices/system/node/node*/distance
echo "== display-class PCI devices and their NUMA node"
for d in /sys/bus/pci/devices/*; do c=$(cat $d/class); case $c in 0x0302*|0x0380*|0x0300*) echo "$(basename $d) class $c vendor $(cat $d/vendor) device $(cat $d/device) numa_node $(cat $d/numa_node) local_cpulist $(cat $d/local_cpulist) speed $(cat $d/current_link_speed 2>/dev/null) width $(cat $d/current_link_width 2>/dev/null)";; esac; done
echo "== kfd topology"
for n in /sys/class/kfd/kfd/topology/nodes/*; do echo "$(basename $n): $(grep -E 'simd_count|location_id|domain|unique_id' $n/properties | tr '\n' ' ')"; for l in $n/io_links/*; do echo "   link $(basename $l): $(grep -E 'type|node_to|weight|min_bandwidth|max_bandwidth' $l/properties | tr '\n' ' ')"; done; done 2>/dev/null | head -60
rocm-smi --showtoponuma 2>/dev/null | head -20
rocm-smi --showbus 2>/dev/null | head
python3 - <<'PY'
import torch
print("torch devices:", torch.cuda.device_count())
p = torch.cuda.get_device_properties(0)
print(p.name, "pci", getattr(p, "pci_bus_id", None), getattr(p, "pci_device_id", None), getattr(p, "pci_domain_id", None))
PY
ulimit -l
cat /proc/meminfo | grep -E 'MemTotal|MemFree|Huge'
