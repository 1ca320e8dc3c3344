// Where does a pinned read-out ring land, and what does it cost when it lands on the other socket?
// For the default hipHostMalloc and for hipHostMallocNumaUser under MPOL_BIND to each NUMA node: the node the pages really are on
// (get_mempolicy MPOL_F_ADDR), device-to-host GB/s of 4K RGB8 frames on two copy streams, with the calling thread pinned to the
// GPU's node and to the other one; then a reader thread (the writer of the export: it touches every byte) on either node.
// build: hipcc --offload-arch=gfx950 -O3 tools/ubench_d2h_numa.hip -o build/ubench_d2h_numa -lpthread
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>
#include <sched.h>
#include <sys/syscall.h>
#include <unistd.h>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)
#define MPOL_DEFAULT 0
#define MPOL_BIND 2
#define MPOL_F_NODE 1
#define MPOL_F_ADDR 2

static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
static int node_of(void* p) { int node = -1; return syscall(SYS_get_mempolicy, &node, nullptr, 0, p, MPOL_F_NODE | MPOL_F_ADDR) == 0 ? node : -1; }
static void bind_memory(int node) {
    unsigned long mask = node < 0 ? 0 : (1ul << node);
    if (syscall(SYS_set_mempolicy, node < 0 ? MPOL_DEFAULT : MPOL_BIND, node < 0 ? nullptr : &mask, node < 0 ? 0 : 64) != 0) perror("set_mempolicy");
}
static std::vector<int> cpus_of(int node) {
    std::vector<int> cpus; char path[128]; snprintf(path, sizeof path, "/sys/devices/system/node/node%d/cpulist", node);
    FILE* f = fopen(path, "r"); if (!f) return cpus;
    char line[4096]; if (fgets(line, sizeof line, f)) for (char* tok = strtok(line, ",\n"); tok; tok = strtok(nullptr, ",\n")) {
        int a, b; if (sscanf(tok, "%d-%d", &a, &b) == 2) for (int c = a; c <= b; c++) cpus.push_back(c); else if (sscanf(tok, "%d", &a) == 1) cpus.push_back(a);
    }
    fclose(f); return cpus;
}
static void pin_to(int node) {
    cpu_set_t set; CPU_ZERO(&set);
    for (int c : cpus_of(node)) CPU_SET(c, &set);
    if (sched_setaffinity(0, sizeof set, &set) != 0) perror("sched_setaffinity");
}

int main() {
    const size_t frame = 3840ull*2160*3; const int frames = 60, slots = 8, rounds = 4;
    char bus[64] = ""; CK(hipDeviceGetPCIBusId(bus, sizeof bus, 0));
    for (char* c = bus; *c; c++) *c = (char)tolower(*c);
    int gpu_node = -1; { std::string p = std::string("/sys/bus/pci/devices/") + bus + "/numa_node"; FILE* f = fopen(p.c_str(), "r"); if (f) { if (fscanf(f, "%d", &gpu_node) != 1) gpu_node = -1; fclose(f); } }
    int nodes = 0; while (!cpus_of(nodes).empty()) nodes++;
    printf("GPU %s on NUMA node %d of %d\n", bus, gpu_node, nodes);
    uint8_t* dev; CK(hipMalloc(&dev, frame*frames)); CK(hipMemset(dev, 1, frame*frames));
    hipStream_t st[2]; for (auto& s : st) CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    for (int mode = -1; mode < nodes; mode++) {
        uint8_t* host;
        if (mode < 0) CK(hipHostMalloc(&host, frame*slots, hipHostMallocDefault));
        else { bind_memory(mode); CK(hipHostMalloc(&host, frame*slots, hipHostMallocNumaUser)); }
        for (size_t i = 0; i < frame*slots; i += 4096) host[i] = 0;
        bind_memory(-1);
        printf("## ring: %s → pages on node %d / %d (first / last page)\n", mode < 0 ? "hipHostMallocDefault" : (std::string("hipHostMallocNumaUser under MPOL_BIND(") + std::to_string(mode) + ")").c_str(), node_of(host), node_of(host + frame*slots - 4096));
        for (int thread_node = 0; thread_node < nodes; thread_node++) {
            pin_to(thread_node);
            CK(hipDeviceSynchronize());
            const double t0 = now();
            for (int r = 0; r < rounds; r++) for (int f = 0; f < frames; f++)
                CK(hipMemcpyAsync(host + (size_t)(f % slots)*frame, dev + (size_t)f*frame, frame, hipMemcpyDeviceToHost, st[f & 1]));
            CK(hipDeviceSynchronize());
            const double dt = now() - t0, bytes = (double)frame*frames*rounds;
            // the writer's side: one thread reads every byte of the ring (what write() into a pipe does)
            const double t1 = now(); uint64_t sum = 0;
            for (int r = 0; r < 2; r++) for (size_t i = 0; i < frame*slots/8; i++) sum += ((const uint64_t*)host)[i];
            const double rd = now() - t1;
            printf("   issuing/reading thread on node %d: D2H %6.2f GB/s = %6.1f frames/s;  one thread reading the ring %5.2f GB/s (%llu)\n", thread_node, bytes/dt/1e9, frames*rounds/dt, 2.0*frame*slots/rd/1e9, (unsigned long long)(sum & 1));
        }
        CK(hipHostFree(host));
    }
    return 0;
}
