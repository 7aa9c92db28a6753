"""CPU-only: the profile post-processing the bench line depends on (tools/prof_summary.py, tools/c4_instep.py) on synthetic rocprofv3
CSVs -- the round-5 review found a 1.22 GB "traffic" figure for a 403 MB kernel that came from averaging two problem sizes of one
(kernel, grid); these tests pin the keying on the size class and the family assignment of the in-step durations."""
import csv
import importlib.util
import json
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _load(name):
    spec = importlib.util.spec_from_file_location(name, os.path.join(ROOT, "tools", name + ".py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def test_pmc_summary_keys_on_the_size_class(tmp_path):
    ps = _load("prof_summary")
    assert ps.size_classes({"a": 10.0, "b": 11.0, "c": 40.0, "d": 42.0, "e": 10.5}) == {"a": 0, "e": 0, "b": 0, "c": 1, "d": 1}
    cols = ["Dispatch_Id", "Kernel_Name", "Grid_Size", "Counter_Name", "Counter_Value", "Start_Timestamp", "End_Timestamp"]

    def write(path, counter, small, big):
        with open(path, "w", newline="") as f:
            w = csv.DictWriter(f, fieldnames=cols)
            w.writeheader()
            t = 0
            for i in range(8):          # one persistent grid for both problem sizes: 8 small launches, then 4 big ones
                w.writerow(dict(Dispatch_Id=i, Kernel_Name="void nnhip::rmsnorm_bwd_rows<64, 2, true>(...)", Grid_Size=131072,
                                Counter_Name=counter, Counter_Value=small, Start_Timestamp=t, End_Timestamp=t + 60000))
                t += 100000
            for i in range(8, 12):
                w.writerow(dict(Dispatch_Id=i, Kernel_Name="void nnhip::rmsnorm_bwd_rows<64, 2, true>(...)", Grid_Size=131072,
                                Counter_Name=counter, Counter_Value=big, Start_Timestamp=t, End_Timestamp=t + 240000))
                t += 300000
    fcsv, wcsv, out = tmp_path / "f.csv", tmp_path / "w.csv", tmp_path / "t.json"
    write(fcsv, "FETCH_SIZE", 131072.0, 524288.0)       # KiB: 2 x 131072 KiB = 268 MB of reads for the small launch
    write(wcsv, "WRITE_SIZE", 131072.0, 524288.0)
    ps.traffic(str(fcsv), str(wcsv), str(out), ["rmsnorm_bwd_rows%0=small", "rmsnorm_bwd_rows%1=big", "rmsnorm_bwd_rows=first"])
    res = json.load(open(out))
    assert res["small"]["hbm_bytes_per_launch"] == (2 * 131072.0 + 131072.0) * 1024
    assert res["big"]["hbm_bytes_per_launch"] == 4 * res["small"]["hbm_bytes_per_launch"]
    assert res["first"]["hbm_bytes_per_launch"] == res["small"]["hbm_bytes_per_launch"] and res["first"]["size_classes_seen"] == 2


def test_c4_instep_assigns_dispatches_to_families(tmp_path, monkeypatch):
    ci = _load("c4_instep")
    cols = ["Kernel_Name", "Start_Timestamp", "End_Timestamp", "Grid_Size_X"]
    rows, t = [], 0

    def k(name, us, grid=131072):
        nonlocal t
        rows.append(dict(Kernel_Name=name, Start_Timestamp=t, End_Timestamp=t + int(us * 1000), Grid_Size_X=grid))
        t += int(us * 1000)
    for step in range(4):               # (the first "step" has no adamw before it and is dropped)
        for _ in range(6):
            k("void nnhip::gemm_pst_kernel<0, true>(nnhip::PstParams)", 188)
            k("void nnhip::gemm_f32_kernel<32, true, true, true, false>(nnhip::GemmParams)", 72)
            k("void nnhip::gemm_pst_kernel<3, true>(nnhip::PstParams)", 259)
            k("void nnhip::gemm_f32_kernel<32, true, true, true, false>(nnhip::GemmParams)", 247)
        k("void nnhip::gemm_pst_kernel<0, true>(nnhip::PstParams)", 1800)
        for _ in range(3):
            k("void nnhip::gemm_f32_group_kernel<32>(nnhip::GemmGroup)", 1450, 393216)
        k("void nnhip::gemm_f32_group_kernel<32>(nnhip::GemmGroup)", 1817, 241664)
        k("nnhip::adamw_multi_kernel(...)", 155)
    src, out = tmp_path / "trace.csv", tmp_path / "instep.json"
    with open(src, "w", newline="") as f:
        w = csv.DictWriter(f, fieldnames=cols)
        w.writeheader()
        w.writerows(rows)
    monkeypatch.setattr("sys.argv", ["c4_instep.py", str(src), str(out), "test"])
    ci.main()
    d = json.load(open(out))
    assert d["_steps"] == 3 and d["_collected"] == "test"
    assert d["qkv_fwd"]["n"] == 18 and abs(d["qkv_fwd"]["us"] - 188) < 1e-6
    assert d["head_fwd"]["n"] == 3 and abs(d["head_fwd"]["us"] - 1800) < 1e-6          # same kernel name as q|k|v: told apart by duration
    assert d["out_fwd"]["n"] == 18 and d["fc2_fwd"]["n"] == 18 and abs(d["fc2_fwd"]["us"] - 247) < 1e-6
    assert d["fc1_fwd_swish"]["n"] == 18
    assert abs(d["layer_dw"]["us"] - 725) < 1e-6 and d["layer_dw"]["units_per_launch"] == 2  # one launch covers two decoder layers
    assert d["head_dw"]["n"] == 3


def test_bench_rccl_choices_and_device_time_guard(tmp_path):
    """bench.py's readers of what it cannot measure itself: RCCL's TUNING lines (algorithm / protocol per collective size) and the
    guard that drops a device-time figure which exceeds the wall time per step (host stalls inside the event pairs)."""
    spec = importlib.util.spec_from_file_location("bench_mod", os.path.join(ROOT, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    log = tmp_path / "rccl.log"
    log.write_text("\n".join([
        "runc:77:77 [0] NCCL INFO RCCL version : 2.26.6-HEAD:64f48b6",
        "runc:77:99 [0] NCCL INFO 128 coll channels, 128 collnet channels, 0 nvls channels, 64 p2p channels, 128 p2p channels per peer",
        "runc:77:99 [0] NCCL INFO AllReduce: 34980448 Bytes -> Algo 1 proto 2 time 412.500000",
        "runc:77:99 [0] NCCL INFO AllReduce: 34980448 Bytes -> Algo 1 proto 2 time 412.500000",
        "runc:77:99 [0] NCCL INFO AllReduce: 16 Bytes -> Algo 0 proto 0 time 9.100000",
        "runc:77:99 [0] NCCL INFO Broadcast: 4 Bytes -> Algo 1 proto 0 time 5.0"]))
    r = bench.rccl_choices(str(log))
    assert "RCCL version" in r["version"] and "coll channels" in r["channels"]
    big = r["choices"][0]
    assert big == {"collective": "AllReduce", "bytes": 34980448, "algorithm": "RING", "protocol": "SIMPLE", "calls": 2}
    assert {"collective": "AllReduce", "bytes": 16, "algorithm": "TREE", "protocol": "LL", "calls": 1} in r["choices"]
    assert "note" in bench.rccl_choices(str(tmp_path / "missing.log")) and "note" in bench.rccl_choices(None)

    class FakeEvent:
        def __init__(self, t):
            self.t = t

        def elapsed_time(self, other):
            return other.t - self.t
    ev = bench.EventTimer()
    for span in (0.020, 0.021, 0.019, 0.500):                # one pair caught a host stall
        ev.pairs.append((FakeEvent(0.0), FakeEvent(span)))
    assert abs(ev.median_ms() - 0.0205) < 1e-9 and ev.mean_ms() > 0.1
    assert abs(bench.device_step_ms(ev, 1, 0.0215) - 0.0205) < 1e-9
    assert bench.device_step_ms(ev, 1, 0.018) is None        # above the wall time per step by more than 5 %: dropped, not reported
    assert bench.device_step_ms(bench.EventTimer(), 1, 1.0) is None
