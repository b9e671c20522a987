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


def _parse_both(tmp_path, text, n, tag):
    """(conversions, stream position, floats) of n fields of `text`: with fscanf("%f,") and with fe_read_frame."""
    exe = tmp_path / "parse_driver"
    if not exe.exists():
        subprocess.run(["gcc", "-O1", "-g", "-std=gnu11", "-ffp-contract=off", "-fsanitize=address,undefined",
                        "-fno-sanitize-recover=undefined", "-o", str(exe), str(ROOT / "tests" / "parse_driver.c"),
                        str(PKG_DIR / "host" / "slam_frontend.c"), "-lm"], check=True)
    src, out = tmp_path / f"{tag}.csv", tmp_path / f"{tag}.bin"
    src.write_text(text)
    subprocess.run([str(exe), str(src), str(n), str(out)], check=True)
    raw = out.read_bytes()
    ok_a, ok_b = struct.unpack_from("<ii", raw, 0)
    pos_a, pos_b = struct.unpack_from("<qq", raw, 8)
    v = np.frombuffer(raw, np.float32, offset=24)
    return (ok_a, pos_a, v[:n]), (ok_b, pos_b, v[n:2 * n])


def test_frame_reader_equals_fscanf_on_every_kind_of_field(tmp_path):
    """fe_read_frame does not go through fscanf (main.c:26-29 does): it must convert the same fields to the same floats and
    leave the stream where fscanf leaves it — the plain decimals "%f" prints (its own fast path: one exact float division),
    values too long or too large for that path, exponents, inf / nan, hexadecimal floats, signs, missing integer or fraction
    parts, white space and line breaks between fields, a missing last comma, end of file inside a frame."""
    rng = np.random.default_rng(11)
    fields = []
    for scale in (1e-6, 1e-3, 1.0, 10.0, 16.0, 17.0, 100.0, 1e4, 1e7, 1e12):
        x = (rng.random(300) * scale).astype(np.float64)
        fields += [f"{v:f}" for v in x[:100]] + [f"{v:.3f}" for v in x[100:150]] + [f"{v:.9g}" for v in x[150:200]]
        fields += [f"{v:e}" for v in x[200:230]] + [f"{-v:f}" for v in x[230:260]] + [f"{v:.12f}" for v in x[260:300]]
    fields += ["16777215.000000", "16777216.000000", "16777217.000000", "16777217.500000", "33554433.000000", "0.000000", "-0.000000",
               "+1.500000", ".5", "5.", "-.25", "007.250000", "1e5", "1E-3", "2.5e+2", "inf", "-inf", "INF", "nan", "-nan", "NaN",
               "infinity", "0x1.8p3", "0X1P-2", "123456789012345678901234567890.5", "0.12345678901234567890123456789",
               "4294967296.000000", "9.999999", "24.000001", "0.023000", "3.4028235e38", "1e39", "1e-46", "1.17549435e-38"]
    # every tie between two neighbouring floats near 1 and near 20, written out in full: the correctly rounded result is the even one
    for base in (1.0, 19.5):
        f0 = np.float32(base)
        for k in range(40):
            f1 = np.nextafter(f0, np.float32(np.inf), dtype=np.float32)
            mid = (float(f0) + float(f1)) / 2.0
            fields += [f"{mid:.30f}", f"{np.nextafter(mid, 0.0):.30f}", f"{np.nextafter(mid, 100.0):.30f}"]
            f0 = f1
    order = rng.permutation(len(fields))
    fields = [fields[i] for i in order]
    seps = [",", ", ", ",\n", ",  ", ",\t", ",\r\n"]   # (white space BEFORE a comma stops the reference's reader for good: below)
    text = "".join(f + seps[i % len(seps)] for i, f in enumerate(fields))
    n = len(fields)
    a, b = _parse_both(tmp_path, text, n, "mixed")
    assert a[0] == n and b[0] == n and a[1] == b[1]
    assert np.array_equal(bits(a[2]), bits(b[2])), np.nonzero(bits(a[2]) != bits(b[2]))[0][:10]
    # no comma behind the last field; fewer fields than asked for (end of file inside the frame): the rest keeps its content
    a, b = _parse_both(tmp_path, "1.250000,2.500000,3.750000", 5, "short")
    assert a[0] == b[0] == 3 and np.array_equal(bits(a[2]), bits(b[2]))
    a, b = _parse_both(tmp_path, "", 3, "empty")
    assert a[0] == b[0] == 0 and np.array_equal(bits(a[2]), bits(b[2]))
    # a field that is no number stops both readers for good: same count, same floats
    a, b = _parse_both(tmp_path, "1.5,2.5,abc,3.5,4.5,", 5, "garbage")
    assert a[0] == b[0] == 2 and np.array_equal(bits(a[2]), bits(b[2]))
    a, b = _parse_both(tmp_path, "1.5 ,2.5,3.5,", 4, "space_before_comma")
    assert a[0] == b[0] == 1 and np.array_equal(bits(a[2]), bits(b[2]))


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
