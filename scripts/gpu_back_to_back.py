#!/usr/bin/env python3
"""Why do launches of the persistent kernel that follow each other without a host round trip run 10-15 % longer each?
Config 2 (IP pattern, 1 GiB synthlog resident in HBM), per-launch kernel time from the library's events (CXG_ASYNC_TIMING=1 for the pending calls), under:
  sync          the synchronous entry (a host round trip of ~15 us between launches)
  sync+idle     the same with 300 us of host sleep between calls
  async8        batches of 8 pending calls on one stream (launches back to back)
  async8+gap    the same with ~25 us of device idle (torch.cuda._sleep on the stream) between the launches
  async8+2out   the same as async8, output arrays alternating
and the shader clock / power the driver reports while each leg runs (sysfs, sampled every 2 ms by a thread).  GPU box only.
Run a second time with CXG_LIB_PATH=coregex_amd/variants/libcoregex_hip_nontrows.so (rows by default-policy stores) for the store-drain hypothesis."""
import glob, os, sys, threading, time
os.environ.setdefault("CXG_ASYNC_TIMING", "1")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import coregex_amd as cx

n = 1 << 30
buf = cx.DeviceBuffer(n)
buf.fill_synth(2, 0xC0FFEE02, 0)
rx = cx.compile(r"\d+\.\d+\.\d+\.\d+")
cnt = rx.find_all_device(buf.ptr, n)
outs = [torch.empty((cnt + 16, 2), dtype=torch.int64, device="cuda") for _ in range(2)]
st = torch.cuda.Stream()


def sysfs(pattern):
    for p in glob.glob(pattern):
        return p
    return None


SCLK = sysfs("/sys/class/drm/card*/device/pp_dpm_sclk")
PWR = sysfs("/sys/class/drm/card*/device/hwmon/hwmon*/power1_average") or sysfs("/sys/class/drm/card*/device/hwmon/hwmon*/power1_input")
FREQ = sysfs("/sys/class/drm/card*/device/hwmon/hwmon*/freq1_input")


class Sampler(threading.Thread):
    def __init__(self):
        super().__init__(daemon=True)
        self.on, self.sclk, self.pwr = True, [], []

    def run(self):
        while self.on:
            try:
                if FREQ:
                    self.sclk.append(int(open(FREQ).read()) / 1e6)
                elif SCLK:
                    for ln in open(SCLK):
                        if "*" in ln:
                            self.sclk.append(float(ln.split(":")[1].strip().split("M")[0]))
                if PWR:
                    self.pwr.append(int(open(PWR).read()) / 1e6)
            except Exception:
                pass
            time.sleep(0.002)


def leg(name, body, reps):
    for _ in range(3):
        body(warm=True)
    torch.cuda.synchronize()
    s = Sampler(); s.start()
    t0 = time.perf_counter()
    ks = []
    for _ in range(reps):
        ks += body(warm=False)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    s.on = False; s.join()
    ks = sorted(ks)
    mean = sum(ks) / len(ks)
    clk = (sum(s.sclk) / len(s.sclk)) if s.sclk else float("nan")
    pw = (sum(s.pwr) / len(s.pwr)) if s.pwr else float("nan")
    print(f"{name:14s} launches {len(ks):4d} kernel ms mean {mean:.4f} median {ks[len(ks)//2]:.4f} min {ks[0]:.4f} max {ks[-1]:.4f}   wall per launch {dt / len(ks) * 1e3:.4f} ms   sclk {clk:7.1f} MHz ({len(s.sclk)} samples)  power {pw:6.1f} W", flush=True)


def sync_body(idle=0.0):
    def f(warm):
        t = cx.Timing()
        ks = []
        for _ in range(8):
            assert rx.find_all_device(buf.ptr, n, outs[0].data_ptr(), cnt + 16, timing=t) == cnt
            ks.append(t.kernel_ms)
            if idle:
                time.sleep(idle)
        return ks
    return f


def async_body(gap=False, two=False):
    def f(warm):
        pend = []
        for i in range(8):
            pend.append(rx.find_all_device_async(buf.ptr, n, outs[i & 1 if two else 0].data_ptr(), cnt + 16, stream=st.cuda_stream))
            if gap:
                with torch.cuda.stream(st):
                    torch.cuda._sleep(60000)
        ks = []
        for p in pend:
            t = cx.Timing()
            assert p.wait(t) == cnt
            ks.append(t.kernel_ms)
        return ks
    return f


print("library", os.environ.get("CXG_LIB_PATH", "product"), "| sysfs:", FREQ or SCLK, PWR)
leg("sync", sync_body(), 16)
leg("sync+idle", sync_body(300e-6), 16)
leg("async8", async_body(), 16)
leg("async8+gap", async_body(gap=True), 16)
leg("async8+2out", async_body(two=True), 16)
leg("sync (again)", sync_body(), 16)
