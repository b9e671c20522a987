"""The product's C front end (host/slam_frontend.c — frame reader, scan clean-up, transform, local map,
rasterisation) checked on the CPU, under AddressSanitizer + UBSan, against the oracle and the reference's
golden vectors.  Reference lines: Subsystem_1/main.c:22-30, 45-118, 155-198, 271-354."""
import struct
import subprocess

import numpy as np

from __graft_entry__ import PKG_DIR, ROOT
from conftest import GOLDEN, bits


def _read_dump(path):
    out, data = {}, open(path, "rb").read()
    pos = 0
    while pos < len(data):
        (ln,) = struct.unpack_from("<I", data, pos); pos += 4
        name = data[pos:pos + ln].decode(); pos += ln
        (nb,) = struct.unpack_from("<Q", data, pos); pos += 8
        out[name] = data[pos:pos + nb]; pos += nb
    return out


def test_frontend_matches_oracle_and_reference_under_sanitizers(orc, golden, tmp_path):
    exe = tmp_path / "frontend_driver"
    subprocess.run(["gcc", "-O1", "-g", "-std=gnu11", "-ffp-contract=off", "-fsanitize=address,undefined",
                    "-fno-sanitize-recover=undefined", "-o", str(exe), str(ROOT / "tests" / "frontend_driver.c"),
                    str(PKG_DIR / "host" / "slam_frontend.c"), "-lm"], check=True)
    subprocess.run([str(exe), str(GOLDEN / "frames_head.csv"), str(tmp_path / "dump.bin")], check=True)
    d = _read_dump(tmp_path / "dump.bin")
    f32 = lambda k: np.frombuffer(d[k], np.float32)
    i32 = lambda k: np.frombuffer(d[k], np.int32)
    # A1/A2 against the reference's golden vectors
    assert i32("got0")[0] == 1079 and i32("eof_got")[0] == 0
    assert np.array_equal(bits(f32("angles")), bits(golden["angles"]))
    assert np.array_equal(bits(f32("ranges0")), bits(golden["ranges_0"]))
    assert np.array_equal(bits(f32("bx0")), bits(golden["scan_x_0"]))
    assert np.array_equal(bits(f32("by0")), bits(golden["scan_y_0"]))
    # A3-A5 on frame 1 against the oracle (itself pinned on the reference)
    ang = orc.beam_angles(-2.351831, 0.004363, 1079)
    lines = (GOLDEN / "frames_head.csv").read_text().splitlines()
    r1 = np.array([float(v) for v in lines[1].split(",")], np.float32)
    x1, y1 = orc.clean_scan(r1, ang)
    tx, ty = orc.transform(x1, y1, [0.15, 0.004, -0.024])
    assert np.array_equal(bits(f32("wx1")), bits(tx)) and np.array_equal(bits(f32("wy1")), bits(ty))
    x0, y0 = orc.clean_scan(golden["ranges_0"], ang)
    mx, my = orc.transform(x0, y0, [0, 0, 0])
    lx, ly = orc.local_map(mx, my, tx, ty, 1.0)
    assert np.array_equal(bits(f32("lx")), bits(lx)) and np.array_equal(bits(f32("ly")), bits(ly))
    assert i32("rc")[0] == 0 and i32("rc1")[0] == 0
    for k, (pix, ld) in enumerate(((0.2, 200), (0.1, 400))):
        grid, m = orc.rasterise(lx, ly, pix, ld)
        rows, cols, gld = i32(f"meta{k}")[:3]
        pixel, minx, miny = np.frombuffer(d[f"meta{k}"], np.float32)[3:6]
        assert (rows, cols, gld) == (m.rows, m.cols, m.ld)
        assert np.array_equal(bits([pixel, minx, miny]), bits([m.pixel, m.min_x, m.min_y]))
        assert np.array_equal(i32(f"grid{k}").reshape(ld, ld), grid)


def test_oracle_whole_program_under_sanitizers(orc, tmp_path):
    """ASan/UBSan build of the CPU restatement over 200 frames (GPU sanitizers are unavailable on the pool)."""
    subprocess.run(["make", "-s", "-C", str(ROOT / "oracle"), "_build/main_cpu_asan"], check=True)
    import json
    info = json.loads((GOLDEN / "datasets.json").read_text())["parity"]
    csv = tmp_path / "p.csv"
    orc.run_tool("gen_dataset", csv, "200", *info["gen_args"][1:])
    r = subprocess.run([str(ROOT / "oracle" / "_build" / "main_cpu_asan"), str(csv), "200", "1079", "1",
                        str(tmp_path / "m.csv")], check=True, capture_output=True, text=True)
    poses = [ln for ln in r.stdout.splitlines() if ln.startswith("pose =")]
    assert poses == (GOLDEN / "parity_pose.txt").read_text().splitlines()[:199]
