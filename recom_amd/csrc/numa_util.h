// numa_util.h — which CPUs sit next to a GPU (Linux sysfs).  Used to keep the request
// stager's pack threads (and the bench's driver thread) on the socket whose memory the
// H2D copy reads: on a 2-socket host, packing on the far socket costs 1.3-2x.
#pragma once
#include <hip/hip_runtime.h>
#include <sched.h>

#include <cctype>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>

namespace fcp {

// CPU set of the NUMA node of `device`; false when the topology cannot be read
// (single node, container without sysfs, ...) — callers then leave affinity alone.
inline bool cpus_near_device(int device, cpu_set_t *set) {
  char bus[64] = {0};
  if (hipDeviceGetPCIBusId(bus, (int)sizeof(bus), device) != hipSuccess) return false;
  for (char *p = bus; *p; ++p) *p = (char)std::tolower((unsigned char)*p);
  char path[256];
  std::snprintf(path, sizeof(path), "/sys/bus/pci/devices/%s/numa_node", bus);
  FILE *f = std::fopen(path, "r");
  if (!f) return false;
  int node = -1;
  const int got = std::fscanf(f, "%d", &node);
  std::fclose(f);
  if (got != 1 || node < 0) return false;
  std::snprintf(path, sizeof(path), "/sys/devices/system/node/node%d/cpulist", node);
  f = std::fopen(path, "r");
  if (!f) return false;
  char list[4096] = {0};
  const bool ok = std::fgets(list, sizeof(list), f) != nullptr;
  std::fclose(f);
  if (!ok) return false;
  CPU_ZERO(set);
  int n = 0;
  for (char *tok = std::strtok(list, ",\n"); tok; tok = std::strtok(nullptr, ",\n")) {
    int lo = 0, hi = 0;
    const int k = std::sscanf(tok, "%d-%d", &lo, &hi);
    if (k == 1) hi = lo;
    if (k < 1) continue;
    for (int c = lo; c <= hi && c < CPU_SETSIZE; ++c) {
      CPU_SET(c, set);
      ++n;
    }
  }
  // only CPUs this process may use anyway
  cpu_set_t allowed;
  if (sched_getaffinity(0, sizeof(allowed), &allowed) == 0) {
    n = 0;
    for (int c = 0; c < CPU_SETSIZE; ++c) {
      if (CPU_ISSET(c, set) && !CPU_ISSET(c, &allowed)) CPU_CLR(c, set);
      if (CPU_ISSET(c, set)) ++n;
    }
  }
  return n > 0;
}

} // namespace fcp
