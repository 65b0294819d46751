"""Lab: where the four stage threads of the sampler's pipeline sit (VV_SAMPLER_PLACE) against the rate of the 8192-item batch.
Reads the SMT sibling of the calling thread's neighbours from sysfs; each placement: 5 samplers, min / median / max."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import videovector_amd as vv
from videovector_amd.synth import SyntheticVideos

def sib(c):
    t = open("/sys/devices/system/cpu/cpu%d/topology/thread_siblings_list" % c).read().strip().replace("-", ",").split(",")
    o = [int(x) for x in t if int(x) != c]
    return o[0] if o else -1

allowed = sorted(os.sched_getaffinity(0))
cur = allowed[len(allowed) // 2] & ~7
full = set(allowed)
c = [cur + i for i in range(8)]
places = {
    "default (common set)": None,
    "four cores": "%d,%d,%d,%d" % (c[1], c[2], c[3], c[4]),
    "walk+stream siblings, negs+frames siblings": "%d,%d,%d,%d" % (c[1], sib(c[1]), c[2], sib(c[2])),
    "walk+stream siblings, negs / frames own cores": "%d,%d,%d,%d" % (c[1], sib(c[1]), c[2], c[3]),
    "walk+negs siblings, stream+frames siblings": "%d,%d,%d,%d" % (c[1], c[2], sib(c[1]), sib(c[2])),
    "walk own core, stream+negs siblings, frames own": "%d,%d,%d,%d" % (c[1], c[2], sib(c[2]), c[3]),
}
ds = SyntheticVideos(seed=1701, n_videos=2048)
print("consumer on cpu %d, sibling of cpu %d = %d" % (cur, c[1], sib(c[1])))
for B in (1024, 8192):
    for name, pl in places.items():
        if pl is None: os.environ.pop("VV_SAMPLER_PLACE", None); os.sched_setaffinity(0, full)
        else: os.environ["VV_SAMPLER_PLACE"] = pl; os.sched_setaffinity(0, {cur})            # the consumer on the group's first core
        ms = []
        for run in range(5):
            s = vv.Sampler(ds.video_id, ds.n_shots, ds.row_base, batch_size=B, context_size=5, num_negative_samples=50, max_buffer_size=5000, negative_swap_percentage=50)
            s.prefetch_start(depth=8, threads=4)
            n = max(8, 400 * 1024 // B)
            for _ in range(4): s.next()
            t0 = time.perf_counter()
            for _ in range(n): s.next()
            ms.append((time.perf_counter() - t0) / n * 1e3)
            s.close()
        a = np.sort(ms)
        print("B %5d  %-50s min %.3f  median %.3f  max %.3f ms per batch" % (B, name, a[0], np.median(a), a[-1]), flush=True)
