"""pytest plugin for debugging a suite that dies without a message: per-test start / finish times, host RSS and device memory,
plus a 5-second sampler thread, appended to gpurun_out/ts.log.   PYTHONPATH=tools python -m pytest -p ts_plugin ..."""
import os
import threading
import time

import psutil

_LOG = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out", "ts.log")
_P = psutil.Process()
_T0 = time.time()


def _line(tag):
    try:
        import torch
        dm = torch.cuda.memory_reserved() / 2**30 if torch.cuda.is_initialized() else 0.0
    except Exception:
        dm = -1.0
    kids = _P.children(recursive=True)
    with open(_LOG, "a") as f:
        f.write("%7.1f s  rss %6.2f GiB  vms %7.1f GiB  torch_reserved %6.2f GiB  threads %3d  children %d  %s\n" % (
            time.time() - _T0, _P.memory_info().rss / 2**30, _P.memory_info().vms / 2**30, dm, _P.num_threads(), len(kids), tag))


def _gpu_busy():
    out = []
    import glob
    for f in sorted(glob.glob("/sys/class/drm/card*/device/gpu_busy_percent")):
        try:
            out.append(open(f).read().strip())
        except OSError:
            out.append("?")
    return ",".join(out)


def _threads():
    rows = []
    pid = os.getpid()
    for t in os.listdir("/proc/%d/task" % pid):
        try:
            st = open("/proc/%d/task/%s/stat" % (pid, t)).read()
            comm = st[st.index("(") + 1:st.rindex(")")]
            f = st[st.rindex(")") + 2:].split()
            state, ut, stt = f[0], int(f[11]), int(f[12])
            wchan = open("/proc/%d/task/%s/wchan" % (pid, t)).read().strip()
            rows.append((ut + stt, comm, state, wchan, t))
        except (OSError, ValueError):
            pass
    rows.sort(reverse=True)
    return "; ".join("%s[%s] %s %s cpu=%d" % (c, t, s, w, u) for u, c, s, w, t in rows[:6])


def _sampler():
    n = 0
    while True:
        time.sleep(5)
        n += 1
        ct = _P.cpu_times()
        _line("(sample)  gpu_busy %s  cpu user %.0f sys %.0f  |  %s" % (_gpu_busy(), ct.user, ct.system, _threads() if n % 4 == 0 else ""))


def pytest_configure(config):
    os.makedirs(os.path.dirname(_LOG), exist_ok=True)
    open(_LOG, "w").close()
    threading.Thread(target=_sampler, daemon=True).start()
    if os.environ.get("TS_CPP_STACKS"):   # C++ backtraces of every thread on a fatal signal
        import torch
        torch._C._set_print_stack_traces_on_fatal_signal(True)


def pytest_runtest_logstart(nodeid, location):
    _line("START " + nodeid)


def pytest_runtest_logfinish(nodeid, location):
    _line("END   " + nodeid)
