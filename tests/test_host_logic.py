"""Host-side logic that needs no GPU: the dirty/progressive bookkeeping of
Connection::update_buffers (connection.rs:370-410), jitter range, strip partition maths,
synthetic input generators."""
import numpy as np

from rsraytracingv3_amd import partition, synth
from rsraytracingv3_amd.connection import Connection, FrameData, RandomTexture
from rsraytracingv3_amd.scene import Scene


class FakeLib:
    def __init__(self, log):
        self.log = log

    def rt_set_random(self, handle, w, h, ptr):
        self.log.append(("rt_set_random", w, h))
        return 0


class FakeDevice:
    """Records the C-ABI calls the host layer would make."""

    def __init__(self):
        self.log = []
        self._lib = FakeLib(self.log)
        self.handle = None
        self.device_id = 0

    def call(self, name, *args):
        self.log.append((name,) + tuple(a if isinstance(a, (int, float, str)) else type(a).__name__ for a in args))


def make_connection(max_objects=None):
    dev = FakeDevice()
    scene = Scene.random_spheres(3, 8, 30, 5, 3, 4)
    conn = Connection(scene, dev, synth.procedural_skybox(16, 8), RandomTexture.start(dev), max_objects=max_objects)
    return dev, scene, conn


def test_update_buffers_dirty_and_progressive_count():
    dev, scene, conn = make_connection()
    assert [c[0] for c in dev.log] == ["rt_set_random", "rt_set_skybox"]
    conn.update_buffers((64, 48), scene)      # last_size starts (0,0) -> dirty (connection.rs:357,373)
    assert conn.frame_data.progressive_count == 0
    sets = [c for c in dev.log if c[0] == "rt_set_scene"]
    assert len(sets) == 1 and sets[0][2] == 80 and sets[0][4] == 2  # 80 object slots, 2 light slots
    conn.update_buffers((64, 48), scene)
    conn.update_buffers((64, 48), scene)
    assert conn.frame_data.progressive_count == 2 and len([c for c in dev.log if c[0] == "rt_set_scene"]) == 1
    scene.camera.fov = 60.0                   # scene != last_scene -> re-serialise, restart accumulation
    conn.update_buffers((64, 48), scene)
    assert conn.frame_data.progressive_count == 0 and len([c for c in dev.log if c[0] == "rt_set_scene"]) == 2
    conn.update_buffers((32, 48), scene)      # size change -> dirty
    assert conn.frame_data.progressive_count == 0 and len([c for c in dev.log if c[0] == "rt_set_scene"]) == 3


def test_live_count_mode_passes_exact_count():
    dev, scene, conn = make_connection(max_objects="live")
    conn.update_buffers((8, 8), scene)
    assert [c for c in dev.log if c[0] == "rt_set_scene"][0][2] == len(scene.objects)


def test_jitter_range_and_determinism():
    dev, scene, conn = make_connection()
    js = []
    for _ in range(200):
        conn.update_buffers((8, 8), scene)
        js.append(conn.frame_data.jitter)
    js = np.array(js)
    assert js.min() >= -0.495 and js.max() < 0.495   # r*0.99 - 0.495 (connection.rs:20,392-399)
    assert js.std() > 0.2
    _, scene2, conn2 = make_connection()
    conn2.update_buffers((8, 8), scene2)
    assert conn2.frame_data.jitter == tuple(js[0])
    conn.update_buffers((8, 8), scene, jitter=(0.125, -0.25))
    assert conn.frame_data.as_bytes()[:8] == FrameData((0.125, -0.25)).as_bytes()[:8]


def test_partition_covers_every_row_exactly_once():
    for height, world, strip in [(1080, 8, 8), (1080, 3, 16), (77, 3, 8), (5, 8, 8), (2160, 8, 8), (100, 1, 8)]:
        seen = np.zeros(height, int)
        for r in range(world):
            idx = partition.local_row_indices(height, r, world, strip)
            assert (np.diff(idx) > 0).all() and idx.size == partition.local_row_count(height, r, world, strip)
            seen[idx] += 1
        assert (seen == 1).all()
        assert partition.max_local_rows(height, world, strip) == max(
            partition.local_row_count(height, r, world, strip) for r in range(world))


def test_assemble_rows_inverts_the_partition():
    h, w, world, strip = 77, 5, 3, 8
    full = np.arange(h * w * 2, dtype=np.float32).reshape(h, w, 2)
    pad = partition.max_local_rows(h, world, strip)
    pieces = []
    for r in range(world):
        idx = partition.local_row_indices(h, r, world, strip)
        piece = np.full((pad, w, 2), -1, np.float32)   # padded like the gather buffers
        piece[: idx.size] = full[idx]
        pieces.append(piece)
    assert np.array_equal(partition.assemble_rows(pieces, h, strip), full)


def test_synthetic_inputs_are_reproducible_and_in_range():
    t = synth.random_texture(5)
    assert t.shape == (600, 600) and t.dtype == np.float32 and t.min() >= 0 and t.max() < 1
    assert np.array_equal(t, synth.random_texture(5)) and not np.array_equal(t, synth.random_texture(6))
    assert (t * (1 << 24) == np.floor(t * (1 << 24))).all()   # 24-bit grid like fastrand/rand f32
    assert t.ravel()[0] == np.float32(synth.SplitMix64(5).f32())
    sky = synth.procedural_skybox(64, 32)
    assert sky.shape == (32, 64, 4) and (sky[..., 3] == 1).all() and sky.min() >= 0 and 20 < sky.max() < 120
    assert sky[2, :, 2].mean() > sky[-3, :, 2].mean()   # blue sky above, dark ground below (row 0 = top)


def test_random_scene_follows_reference_distributions():
    s = Scene.random_spheres(3.0, 8.0, 50.0, 100, 42)
    spheres = [o for o in s.objects if o.geometry.kind == 0]
    assert s.objects[-1].geometry.kind == 1 and s.objects[-1].geometry.size == 100_000.0   # plane pushed last
    assert 50 <= len(spheres) <= 100
    for o in spheres:
        r = o.geometry.radius
        c = o.geometry.center
        assert 3.0 <= r <= 8.0 and c[1] == r and np.hypot(c[0], c[2]) <= 50.0 + 1e-4
    for i, a in enumerate(spheres):    # no overlaps (scene.rs:171-187)
        for b in spheres[:i]:
            d = np.linalg.norm(np.subtract(a.geometry.center, b.geometry.center))
            assert d >= a.geometry.radius + b.geometry.radius - 1e-4
    rough0 = np.mean([o.material.roughness == 0.0 for o in spheres])
    emissive = np.mean([o.material.emission_strength > 0 for o in spheres])
    assert 0.05 < rough0 < 0.45 and 0.02 < emissive < 0.4
    assert s.camera.position == (55.0, 65.0, 55.0) and s.reflection_limit == 3


def test_trace_union_counts_shared_device_time_once(tmp_path):
    """tools/trace_union.py: two overlapping dispatches and one apart -> union of the intervals, not their sum."""
    import os
    import subprocess
    import sys
    d = tmp_path / "stats"
    d.mkdir()
    rows = ['"Kind","Agent_Id","Queue_Id","Stream_Id","Thread_Id","Dispatch_Id","Kernel_Id","Kernel_Name","Correlation_Id",'
            '"Start_Timestamp","End_Timestamp"']
    for k, (s, e, name) in enumerate([(1000, 3000, "render_blocks<2, 7, true>"), (2000, 4000, "render_blocks<2, 7, true>"),
                                      (10000, 11000, "render_blocks<2, 7, true>"), (0, 99999, "resolve_batch_kernel")]):
        rows.append('"KERNEL_DISPATCH","Agent 2",1,0,1,%d,1,"%s",%d,%d,%d' % (k, name, k, s, e))
    (d / "x_kernel_trace.csv").write_text("\n".join(rows) + "\n")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = subprocess.run([sys.executable, os.path.join(root, "tools", "trace_union.py"), str(d)], capture_output=True, text=True,
                         check=True).stdout
    assert "last 3 launches (1 kernel(s) each)" in out
    assert "mean kernel time per launch   0.0017 ms" in out          # (2000 + 2000 + 1000) / 3 ns
    assert "device time per launch        0.0013 ms" in out          # union 3000 + 1000 = 4000 / 3 ns
    assert "kernels running at once       1.25" in out
    # a launch whose render step is two kernels (render_blocks, then tail_stream): both count towards the launch
    rows = rows[:1]
    for k, (s, e, name) in enumerate([(1000, 3000, "render_blocks<2, 7, true, true>"), (3100, 3600, "tail_stream<2, 5>"),
                                      (5000, 7000, "render_blocks<2, 7, true, true>"), (7100, 7600, "tail_stream<2, 5>")]):
        rows.append('"KERNEL_DISPATCH","Agent 2",1,0,1,%d,1,"%s",%d,%d,%d' % (k, name, k, s, e))
    (d / "x_kernel_trace.csv").write_text("\n".join(rows) + "\n")
    out = subprocess.run([sys.executable, os.path.join(root, "tools", "trace_union.py"), str(d)], capture_output=True, text=True,
                         check=True).stdout
    assert "last 2 launches (2 kernel(s) each)" in out
    assert "mean kernel time per launch   0.0025 ms" in out          # (2000 + 500) ns per launch
    assert "kernels running at once       1.00" in out


def test_pmc_summary_adds_the_two_render_kernels_of_a_launch(tmp_path):
    """tools/pmc_summary.py: a launch's render step is render_blocks followed by tail_stream (RT_OPT_TAIL_SPILL); the
    per-launch figures bench.py imports are the SUM of the two kernels' per-dispatch means, the parts stay readable."""
    import json
    import os
    import subprocess
    import sys
    d = tmp_path / "pmc1"
    d.mkdir()
    head = ('"Correlation_Id","Dispatch_Id","Agent_Id","Queue_Id","Process_Id","Thread_Id","Grid_Size","Kernel_Id","Kernel_Name",'
            '"Workgroup_Size","LDS_Block_Size","Scratch_Size","VGPR_Count","Accum_VGPR_Count","SGPR_Count","Counter_Name",'
            '"Counter_Value","Start_Timestamp","End_Timestamp"')
    rows = [head]
    names = {"blocks": "void rt355::(anonymous namespace)::render_blocks<2, 7, true, true>(rt355::DeviceScene, rt355::FrameParams)",
             "tail": "void rt355::(anonymous namespace)::tail_stream<2, 5>(rt355::DeviceScene, rt355::FrameParams, unsigned int, unsigned int)",
             "resolve": "rt355::(anonymous namespace)::resolve_batch_kernel(rt355::ResolveParams)"}
    k = 0
    for launch in range(8):
        for kind, counters in (("blocks", {"FETCH_SIZE": 1000.0, "SQ_INSTS_VALU": 400.0, "GRBM_GUI_ACTIVE": 8000.0}),
                               ("tail", {"FETCH_SIZE": 250.0, "SQ_INSTS_VALU": 100.0, "GRBM_GUI_ACTIVE": 2000.0}),
                               ("resolve", {"FETCH_SIZE": 100.0})):
            for c, v in counters.items():
                k += 1
                rows.append('%d,%d,"Agent 4",1,1,1,64,1,"%s",128,0,0,72,0,96,"%s",%f,0,1' % (k, k, names[kind], c, v))
    (d / "x_counter_collection.csv").write_text("\n".join(rows) + "\n")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = tmp_path / "traffic.json"
    subprocess.run([sys.executable, os.path.join(root, "tools", "pmc_summary.py"), str(d), "--json", str(out)], check=True,
                   capture_output=True, text=True)
    t = json.loads(out.read_text())
    r = t["render_kernel"]
    assert "render_blocks<2, 7, true, true>" in r["kernel"] and "tail_stream<2, 5>" in r["kernel"]
    assert r["counters_per_launch"]["FETCH_SIZE"] == 1250.0 and r["fetch_bytes"] == 1250.0 * 1024
    assert abs(r["valu_issue_util"] - 500.0 * 2.0 / (1024.0 * 10000.0 / 8.0)) < 1e-4   # over the busy cycles of both kernels
    assert set(t["render_parts"]) == {"render_blocks<2, 7, true, true>", "tail_stream<2, 5>"}
    assert t["resolve_kernel"]["fetch_bytes"] == 2 * 100.0 * 1024                      # coalesced streams: FETCH_SIZE doubled
    assert len(t["kernel_source_sha256"]) == 64


def test_committed_bench_line_keeps_the_contract():
    """The JSON line bench.py printed on the GPU box (profiles/r04_bench_driver_flags.json, the driver's flags): every key
    the bench contract names is there, with the types and relations it asks for; the roofline is the binding one (the FP32 vector
    ALU), its fraction is only printed while it is one, the hardware's view stands next to it; the counters imported from a committed
    PMC profile name their source."""
    import json
    import os
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    with open(os.path.join(root, "profiles", "r04_bench_driver_flags.json")) as f:
        d = json.loads([line for line in f if line.startswith("{")][-1])
    for key, kind in (("metric", str), ("value", float), ("unit", str), ("n_gpus", int), ("steps", int), ("warmup", int),
                      ("ms_per_step", float), ("higher_is_better", bool), ("scaling", str), ("dtype", str), ("data", str),
                      ("config", dict), ("roofline", dict), ("cpu_baseline", dict)):
        assert isinstance(d[key], kind), key
    assert d["vs_baseline"] is None and d["unit"] == "Mrays/s" and d["dtype"] == "f32" and d["data"] == "synthetic"
    assert d["steps"] == 20 and d["warmup"] == 5 and d["n_gpus"] == 1 and d["higher_is_better"] is True
    assert "workload" in d["config"] and "model" not in d["config"]
    r = d["roofline"]
    assert r["bound"] == "valu" and r["unit"] == "TFLOP/s" and r["peak"] == 157.3
    assert abs(r["frac_alg_reference_units_may_exceed_1"] - r["achieved"] / r["peak"]) < 1e-4
    assert r["frac"] == r["frac_alg_reference_units_may_exceed_1"] and 0.0 < r["frac"] < 1.0   # printed as `frac` while it is a fraction
    assert 0.3 < r["issue_util"] < 1.0 and "reference" in r["achieved_basis"] and "direction-mask table" in r["achieved_basis"]
    assert r["traffic"] > 0
    h = r["hbm"]
    assert h["unit"] == "GB/s" and h["peak"] == 8000.0 and abs(h["frac"] - h["achieved"] / h["peak"]) < 1e-4
    assert "algorithmic" in h["achieved_basis"] and abs(h["measured_gbs"] - r["traffic"] / (r["avg_launch_ms"] * 1e-3) / 1e9) < 1.0
    # launches in flight (the library's rule: four of this size): the device time per launch (event span) is below one kernel pair's own duration
    assert r["launches_in_flight"] == 4 and r["avg_launch_ms"] < r["kernel_avg_duration_ms"]
    # value = segments / time: consistent with ms_per_step and the segments per pixel the line reports
    px = 1920 * 1080
    # a step is one launch of config.frames_per_step frames (8 per GPU)
    per_step = d["config"]["frames_per_step"]
    assert per_step == 8 * d["n_gpus"] and d["scaling"] == "weak"
    assert abs(d["value"] - d["config"]["segments_per_pixel"] * px * per_step / (d["ms_per_step"] * 1e-3) / 1e6) / d["value"] < 1e-3
    assert d["config"]["block_of_frames"]["frames_per_block"] == d["steps"]  # the protocol of rounds 1-2, still reported
    c = d["cpu_baseline"]
    assert c["kind"] in ("port", "reference") and c["cores"] >= 1 and c["value"] > 0 and c["unit"] == "Mrays/s" and c["sample"]
    assert c["host_cores"] >= c["cores"]
    assert d["valu"]["frac"] == r["frac"] and d["valu"]["peak"] == 157.3   # (the alias older tools read)
    # imported, not measured in the run: says so, and from which kernel sources
    src = r["traffic_source"]
    assert src["file"] == "profiles/traffic_r04.json" and src["used"] is True and len(src["kernel_source_sha256"]) == 16
    for leg in ("one_frame_per_launch", "reference_80_slots"):
        assert d["config"][leg]["mrays_per_s"] > 0
    assert d["config"]["one_frame_per_launch"]["frames_per_block"] == 160   # the frames of a headline block, one per launch


def test_a_reference_unit_fraction_above_one_is_never_printed_as_a_fraction():
    """Config 3 (1 024 spheres): the launches test four of the 1 025 objects the reference's scan would, so F_alg / t is several
    times the VALU peak — the line prints it under a key that says so, `frac` is null, `issue_util` carries the hardware's view."""
    import json
    import os
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    with open(os.path.join(root, "profiles", "r04_bench_config3.json")) as f:
        d = json.loads([line for line in f if line.startswith("{")][-1])
    r = d["roofline"]
    assert "1024 spheres" in d["config"]["workload"] and r["bound"] == "valu"
    assert r["frac"] is None and r["frac_alg_reference_units_may_exceed_1"] > 1.0 and d["valu"]["frac"] is None
    assert 0.2 < r["issue_util"] < 1.0 and "candidate lists" in r["achieved_basis"]
    assert r["traffic_source"]["file"] == "profiles/traffic_r04_config3.json" and r["traffic"] > r["hbm"]["alg_bytes_per_launch"] * 0.5
    assert d["value"] >= 38000.0   # the round's target for this configuration


def test_profile_counters_are_only_quoted_for_the_sources_they_were_measured_on():
    """profiles/traffic_r04*.json record the sha256 of the kernel sources they were collected on; bench.py quotes them only when
    the tree still has those sources.  The committed profiles must belong to the committed sources."""
    import json
    import os
    from rsraytracingv3_amd import _native
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    for name in ("traffic_r04.json", "traffic_r04_config3.json"):
        with open(os.path.join(root, "profiles", name)) as f:
            prof = json.load(f)
        assert prof["kernel_source_sha256"] == _native.kernel_source_hash(), \
            "csrc/ changed since %s was collected: re-run tools/collect_round4.sh (bench.py now reports traffic null)" % name
