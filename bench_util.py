"""bench_util.py — what bench.py's legs share and what does not depend on the workload: the whole-stream parity comparison, provenance of
the counter passes, sensors, GPU count without touching HIP, the queue-placement decision.  Nothing here imports torch at module level or
makes a HIP call by being imported (bench.py's launcher imports it before it starts its children)."""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))

HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: HBM3E 8 TB/s spec (6.3 TB/s measured copy)
FPGA_MPIXELS = 268.0           # README.md:22, Kintex-7

GOP_CODE, END_CODE = b"\x00\x00\x01\xb8", b"\x00\x00\x01\xb7"


# ---------------------------------------------------------------------------------------------------------------------
# provenance: which tree the committed counter passes measured
# ---------------------------------------------------------------------------------------------------------------------
def kernel_source_sha(path):
    """sha256 of the kernel source as the compiler sees it: comments dropped, runs of white space collapsed - a reworded comment does
    not make the counter passes stale (tools/make_pmc_traffic.py computes the same)"""
    import hashlib
    import re
    text = open(path, encoding="utf-8").read()
    text = re.sub(r"/\*.*?\*/", " ", text, flags=re.S)
    text = re.sub(r"//[^\n]*", " ", text)
    return hashlib.sha256(" ".join(text.split()).encode()).hexdigest()


def source_shas():
    """What the running tree is: git HEAD (if this is a checkout) and the sha256 of the kernel source (kernel_source_sha).  profiles/pmc_traffic.json
    carries the same two values for the tree its PMC passes ran on (tools/profile_round.sh)."""
    import subprocess
    ksha = kernel_source_sha(os.path.join(ROOT, "fpga-mpeg2-encoder_amd", "csrc", "m2v_kernels.hpp"))
    try:
        head = subprocess.run(["git", "-C", ROOT, "rev-parse", "HEAD"], capture_output=True, text=True, timeout=10).stdout.strip() or None
    except Exception:  # noqa: BLE001
        head = None
    return head, ksha


def pmc_traffic(key):
    """roofline.traffic: HBM bytes per launch of the dominant kernel from the PMC passes (FETCH_SIZE x 2 + WRITE_SIZE,
    MI355X_MICROARCH.md), collected by tools/profile_round.sh in separate rocprofv3 runs of this same command and kept in
    profiles/pmc_traffic.json together with the tree they measured.  `traffic_stale` says whether the kernel source has changed
    since (a counter pass cannot run inside the timed job: it serialises the dispatches)."""
    tpath = os.path.join(ROOT, "profiles", "pmc_traffic.json")
    out = {"traffic": None}
    try:
        t = json.load(open(tpath))
    except Exception:  # noqa: BLE001
        return out
    if t.get(key) is None:
        return out
    head, ksha = source_shas()
    out.update({"traffic": t[key], "traffic_source": "profiles/pmc_traffic.json (separate --pmc passes of this command)",
                "traffic_measured_at": {"head": t.get("head"), "kernel_sha": t.get("kernel_sha")},
                "running": {"head": head, "kernel_sha": ksha},
                "traffic_stale": t.get("kernel_sha") != ksha})
    return out


def pmc_valu_busy(key):
    """roofline.valu_busy: the fraction of the dispatch's SIMD-cycles in which the vector ALU was executing (4 x SQ_ACTIVE_INST_VALU /
    (1024 x GRBM_GUI_ACTIVE / 8); `valu_busy_while_resident`: of the cycles the shader engines held waves, SQ_BUSY_CYCLES / 32), computed
    by tools/make_pmc_traffic.py from the committed SQ counter pass and kept in profiles/pmc_traffic.json under `valu_busy`: which roof
    binds the dominant kernel is this number, not the HBM fraction."""
    try:
        t = json.load(open(os.path.join(ROOT, "profiles", "pmc_traffic.json")))
    except Exception:  # noqa: BLE001
        return {}
    v = (t.get("valu_busy") or {}).get(key)
    if v is None:
        return {}
    _, ksha = source_shas()
    out = {"valu_busy": v, "valu_busy_source": t["valu_busy"].get("source"),
           "valu_busy_stale": t["valu_busy"].get("kernel_sha", t.get("kernel_sha")) != ksha}
    if t["valu_busy"].get(key + "_while_resident") is not None:
        out["valu_busy_while_resident"] = t["valu_busy"][key + "_while_resident"]
    return out


# ---------------------------------------------------------------------------------------------------------------------
# whole-stream parity: the GPU stream of a multi-GOP sequence against the oracle's streams of its GOPs
# ---------------------------------------------------------------------------------------------------------------------
def gop_time_code(n):
    """bytes 4..7 of a group_of_pictures_header for sequence frame number n (24 fps time code, closed_gop = 1,
    RTL:2645-2656, 2685-2698): the one field of a GOP that depends on where it sits in the sequence"""
    hh = min(n // 86400, 63)
    return ((hh << 26) | (((n // 1440) % 60) << 20) | (1 << 19) | (((n // 24) % 60) << 13) | ((n % 24) << 7) | (2 << 5)).to_bytes(4, "big")


def split_gops(data):
    """-> (bytes before the first GOP header, [bytes of each GOP], bytes from the sequence end code on)"""
    idx, pos = [], data.find(GOP_CODE)
    while pos >= 0:
        idx.append(pos)
        pos = data.find(GOP_CODE, pos + 4)
    end = data.rfind(END_CODE)
    return data[:idx[0]], [data[a:b] for a, b in zip(idx, idx[1:] + [end])], data[end:]


def compare_with_per_gop_oracle(gpu_stream_bytes, oracle_gop_streams, gop):
    """The GPU stream of a multi-GOP sequence against the oracle's streams of its GOPs, each encoded as a sequence of
    its own (closed GOPs): sequence headers, every GOP (header, time code computed here, all pictures) and the end
    code + final-word padding.  -> list of problems (empty = byte-identical)"""
    head, gops, tail = split_gops(gpu_stream_bytes)
    bad = []
    if len(gops) != len(oracle_gop_streams):
        bad.append("GPU stream has %d GOPs, expected %d" % (len(gops), len(oracle_gop_streams)))
    for k, ref in enumerate(oracle_gop_streams[:len(gops)]):
        rhead, rgops, _ = split_gops(ref)
        if k == 0 and head != rhead:
            bad.append("sequence headers differ")
        if gops[k][:4] != GOP_CODE or gops[k][4:8] != gop_time_code(k * gop):
            bad.append("GOP %d: header / time code" % k)
        if len(rgops) != 1 or gops[k][8:] != rgops[0][8:]:
            bad.append("GOP %d: pictures differ from the oracle" % k)
    body = len(gpu_stream_bytes) - len(tail)
    want_total = ((body + 4) // 32 + 1) * 32                      # end code, then the final 32-byte word always leaves (RTL:2932-2937)
    if tail != END_CODE + bytes(want_total - body - 4):
        bad.append("end code / final padding")
    return bad


def rtl_sim_probe():
    """the RTL under a Verilog simulator is the parity oracle and CPU baseline the metric names.  It
    runs wherever `iverilog` + `vvp` are installed and M2V_RTL points at mpeg2encoder.v (tools/run_rtl_oracle.py); this
    image and the GPU box have neither, so the line says so instead of pretending."""
    import shutil
    iv, vvp, ver, rtl = shutil.which("iverilog"), shutil.which("vvp"), shutil.which("verilator"), os.environ.get("M2V_RTL")
    if not (((iv and vvp) or ver) and rtl and os.path.exists(rtl)):
        return {"available": False, "iverilog": iv, "vvp": vvp, "verilator": ver, "rtl": rtl,
                "note": "RTL oracle unavailable: no Verilog simulator / RTL file on this host; parity is against oracle/m2v_oracle.c "
                        "(line-cited C restatement of the RTL, parity unpinned by the reference - DESIGN.md section 5)"}
    import subprocess
    t0 = time.perf_counter()
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "run_rtl_oracle.py"), "--rtl", rtl], capture_output=True, text=True)
    lines = r.stdout.strip().splitlines()
    try:
        verdict = json.loads(lines[-1])           # the tool's last line: RTL vs oracle vs product (m2v_tb), known answers, simulator
    except (ValueError, IndexError):
        verdict = {"available": True}
    # the RTL against the ORACLE is what pins parity; the tool's exit code also covers RTL against the product and the known answers
    verdict.update({"identical_to_oracle": bool(verdict.get("rtl_equals_oracle")) if "rtl_equals_oracle" in verdict else None,
                    "tool_exit_code": r.returncode, "seconds": round(time.perf_counter() - t0, 1), "cores": 1, "log": lines[-10:-1]})
    return verdict


# ---------------------------------------------------------------------------------------------------------------------
# sensors: the amdgpu sysfs files of the card the rank's HIP device IS
# ---------------------------------------------------------------------------------------------------------------------
def sysfs_card_of(pci_bus_id, drm_root="/sys/class/drm"):
    """The /sys/class/drm/cardN/device directory whose PCI address is `pci_bus_id` ("0000:c1:00.0", what m2v_device_pci_bus_id gives for
    the HIP device a rank runs on), or None.  By ADDRESS, never by position in the list: a lease that shows ONE of a node's eight GPUs to
    HIP still shows all eight cards in sysfs, and card 0 is then somebody else's idle GPU (round 5's sensors read exactly that)."""
    import glob
    if not pci_bus_id:
        return None
    want = pci_bus_id.strip().lower()
    for d in sorted(glob.glob(os.path.join(drm_root, "card[0-9]*", "device"))):
        if "-" in os.path.basename(os.path.dirname(d)):       # connectors (card0-DP-1) are not cards
            continue
        if os.path.basename(os.path.realpath(d)).lower() == want:
            return d
    return None


def gpu_sensors(card_dir):
    """Clocks / power / temperature as amdgpu's sysfs files under `card_dir` (sysfs_card_of) give them (no subprocess, no SMI library:
    readable by an ordinary user where the files exist at all); None for what cannot be read, None altogether without a card."""
    import glob
    if not card_dir or not os.path.isdir(card_dir):
        return None
    d = card_dir

    def current(name):                       # "1: 2400Mhz *" marks the level in use
        try:
            for ln in open(os.path.join(d, name)):
                if ln.rstrip().endswith("*"):
                    return int("".join(ch for ch in ln.split(":")[1] if ch.isdigit()))
        except (OSError, ValueError, IndexError):
            pass
        return None

    def hwmon(name, scale):
        for f in glob.glob(os.path.join(d, "hwmon", "hwmon*", name)):
            try:
                return round(int(open(f).read().strip()) * scale, 1)
            except (OSError, ValueError):
                pass
        return None
    out = {"sclk_mhz": current("pp_dpm_sclk"), "mclk_mhz": current("pp_dpm_mclk"),
           "power_w": hwmon("power1_average", 1e-6) or hwmon("power1_input", 1e-6), "temp_c": hwmon("temp1_input", 1e-3)}
    return out if any(v is not None for v in out.values()) else None


SENSOR_MIN_BUSY_SCLK_MHZ = 1000          # a GPU that delivers the loop's rate is not at its idle clock


def sensors_verdict(samples):
    """-> (plausible, reason).  The samples were taken WHILE the loop ran: a mean shader clock under 1 GHz, or a power reading that never
    moves, is the signature of a card that is not the one doing the work (or of files that do not report) - then the line carries null
    sensors and this reason instead of numbers nobody should draw conclusions from."""
    clk = [s["sclk_mhz"] for s in samples if s.get("sclk_mhz") is not None]
    pw = [s["power_w"] for s in samples if s.get("power_w") is not None]
    if not clk and not pw:
        return False, "no sensor file readable"
    if clk and sum(clk) / len(clk) < SENSOR_MIN_BUSY_SCLK_MHZ:
        return False, "mean sclk %.0f MHz during the loop: an idle card's clock, not this GPU's" % (sum(clk) / len(clk))
    if len(pw) >= 8 and max(pw) == min(pw):
        return False, "power reading constant (%.1f W in %d samples): not a live sensor" % (pw[0], len(pw))
    return True, None


def visible_gpus():
    """How many GPUs a rank of this job would see, WITHOUT touching the HIP runtime (the launcher must not initialise the GPU
    before it starts its children): the *_VISIBLE_DEVICES list if one is set, else the KFD topology's nodes that have SIMDs.
    None when neither can be read (then the ranks themselves check, as before)."""
    for var in ("HIP_VISIBLE_DEVICES", "ROCR_VISIBLE_DEVICES", "CUDA_VISIBLE_DEVICES"):
        v = os.environ.get(var)
        if v is not None:
            return len([x for x in v.split(",") if x.strip() != ""])
    import glob
    n, seen = 0, False
    for f in glob.glob("/sys/class/kfd/kfd/topology/nodes/*/properties"):
        try:
            for ln in open(f):
                k, _, val = ln.partition(" ")
                if k == "simd_count":
                    seen = True
                    n += int(val) > 0
        except (OSError, ValueError):
            pass
    return n if seen else None


def hbm_copy_rate(torch, dev):
    """Achievable HBM bandwidth of this device with a plain device-to-device copy (read + write bytes), GB/s."""
    n = 1 << 30
    a = torch.empty(n, dtype=torch.uint8, device=dev)
    b = torch.empty_like(a)
    b.copy_(a)
    torch.cuda.synchronize(dev)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10):
        b.copy_(a)
    e1.record()
    torch.cuda.synchronize(dev)
    return 10 * 2.0 * n / (e0.elapsed_time(e1) * 1e-3) * 1e-9


# ---------------------------------------------------------------------------------------------------------------------
# Queue placement of two handles' streams.
# How the K sequences are submitted is settled in the warm-up: two handles taking turns (m2v_encode_resident_begin / _end), or one blocking
# call after the other.  Taking turns wins by ~8 % when the handles' streams sit on different hardware queues of the HIP runtime; whether
# they do is the runtime's choice, and one box in ten puts two streams of a process on ONE queue - the sequences then run one after the
# other (profiles/r04_queue_ab.txt reproduces it with GPU_MAX_HW_QUEUES=1).  So the placement is probed, untimed, and repaired: a handle
# whose sequences do not overlap with the other one's gets a NEW stream (the runtime deals its streams to the queues in turn), up to
# PLACEMENT_MAX_NEW_STREAMS times; then a stream of another PRIORITY - those never share a queue with default-priority ones (overlap
# guaranteed, ~3 % behind the best placement).  Blocking calls are the last resort.  The thresholds live here, once.
# ---------------------------------------------------------------------------------------------------------------------
PLACEMENT_OVERLAP_GAIN = 0.96            # in flight must beat the one-stream form by 4 % to count as overlapping
PLACEMENT_BEATS_BLOCKING = 0.985         # ... or the blocking form by 1.5 %: then there is nothing to repair, whatever the streams sit on
PLACEMENT_MAX_NEW_STREAMS = 3


def placement_next_action(t_in_flight, t_one_stream, new_streams, priority_tried, repair_allowed=True, t_blocking=None):
    """One row of the decision table: given the probe times (seconds for the same number of sequences) -> "keep" (the sequences overlap,
    or nothing more can be tried), "new_stream" (give the last handle a fresh stream and probe again) or "priority" (give it a stream of
    another priority and probe again).  A GPU at its power limit gains little from ANY overlap (profiles/r06_experiments.txt item 23:
    blocking 1.021 ms, one stream 1.024, in flight 0.993 - and 1.03 - 1.05 after the re-streamings): when in flight already beats the
    blocking form the placement is left alone - a re-streaming cannot be taken back."""
    if t_in_flight <= PLACEMENT_OVERLAP_GAIN * t_one_stream or not repair_allowed:
        return "keep"
    if t_blocking is not None and t_in_flight <= PLACEMENT_BEATS_BLOCKING * t_blocking:
        return "keep"
    if new_streams < PLACEMENT_MAX_NEW_STREAMS:
        return "new_stream"
    if not priority_tried:
        return "priority"
    return "keep"


def placement_submission(t_in_flight, t_blocking):
    """... and once the repair is over: the form the timed loop uses"""
    return "blocking" if t_blocking < t_in_flight else "in_flight"


def settle_queue_placement(probe_in_flight, probe_blocking, probe_one_stream, restream, repair_allowed=True, per_step=1.0):
    """Runs the probe / repair loop.  probe_*: callables -> seconds for the probe's sequences in that form; restream(kind): "new_stream" /
    "priority" applied to the last handle.  -> (submission, placement record for the bench line)."""
    t_sync = probe_blocking()
    t_serial = probe_one_stream()
    t_fly = probe_in_flight()
    rec = {"new_streams": 0, "priority": 0,
           "probe_ms_per_step": {"blocking": round(t_sync * per_step * 1e3, 3), "one_stream": round(t_serial * per_step * 1e3, 3),
                                 "in_flight": [round(t_fly * per_step * 1e3, 3)]},
           "thresholds": {"overlap_gain": PLACEMENT_OVERLAP_GAIN, "beats_blocking": PLACEMENT_BEATS_BLOCKING, "max_new_streams": PLACEMENT_MAX_NEW_STREAMS}}
    while True:
        act = placement_next_action(t_fly, t_serial, rec["new_streams"], rec["priority"] != 0, repair_allowed, t_sync)
        if act == "keep":
            break
        restream(act)
        if act == "new_stream":
            rec["new_streams"] += 1
        else:
            rec["priority"] = 1
        t_fly = probe_in_flight()
        rec["probe_ms_per_step"]["in_flight"].append(round(t_fly * per_step * 1e3, 3))
    return placement_submission(t_fly, t_sync), rec
