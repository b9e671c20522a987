"""HTTP/JSON scan transport compatible with the reference's simulator
(Submodule_2/Lidar_server.py:8-33) and edge client (Submodule_2/esp32_edge.c:52-99)."""
import http.client
import json
import re

import numpy as np
import pytest

from __graft_entry__ import load_package
from conftest import GOLDEN, bits


def _wire():
    load_package()
    import importlib
    return importlib.import_module("hardware_acceleration_of_lidar_slam_amd.lidar_wire")


def test_server_and_client_speak_the_reference_format(golden):
    wire = _wire()
    srv = wire.ScanServer(GOLDEN / "frames_head.csv")            # pandas semantics: line 0 is the header
    try:
        # raw exchange, as a foreign client would see it: one chunk holding a JSON array
        c = http.client.HTTPConnection("127.0.0.1", srv.port, timeout=10)
        c.request("GET", "/?param=&param=0&pose_x=0.250000&pose_y=-1.500000")   # esp32_edge.c:59
        r = c.getresponse()
        assert r.status == 200 and r.getheader("Content-type") == "application/json"
        assert r.getheader("Transfer-Encoding") == "chunked"
        arr = json.loads(r.read())
        assert isinstance(arr, list) and len(arr) == 1079
        c.close()
        assert srv.requests[-1] == (0, "0.250000", "-1.500000")       # the pose rides along (ESP32_Server.c:45-54)
        # param = N serves CSV line N + 1; values survive the double -> float conversion of the client bit for bit
        lines = (GOLDEN / "frames_head.csv").read_text().splitlines()
        for row in (0, 1):
            got = wire.fetch_scan("127.0.0.1", srv.port, row, 1.0, 2.0)
            want = np.array([np.float32(v) for v in lines[row + 1].split(",")], np.float32)
            assert np.array_equal(bits(got), bits(want))
    finally:
        srv.close()
    # without the pandas header quirk, row 0 is the first frame = what the reference's own CSV reader parses
    srv = wire.ScanServer(GOLDEN / "frames_head.csv", pandas_header=False)
    try:
        assert np.array_equal(bits(wire.fetch_scan("127.0.0.1", srv.port, 0)), bits(golden["ranges_0"]))
    finally:
        srv.close()


@pytest.mark.gpu
def test_mapper_fed_over_http_matches_reference_poses(orc, tmp_path):
    """Frames fetched over the reference's wire format and fed to the device-resident frame loop give the
    reference's pose log (first 60 frames of the parity set)."""
    wire = _wire()
    pkg = load_package()
    info = json.loads((GOLDEN / "datasets.json").read_text())["parity"]
    csv = tmp_path / "p.csv"
    orc.run_tool("gen_dataset", csv, "60", *info["gen_args"][1:])
    srv = wire.ScanServer(csv, pandas_header=False)
    eng = pkg.Engine(0)
    try:
        poses = wire.run_mapper_over_http(pkg, eng, "127.0.0.1", srv.port, 60)
    finally:
        eng.close()
        srv.close()
    want = [[float(v) for v in re.split(r"\s+", ln.split("=")[1].strip())] for ln in
            (GOLDEN / "parity_pose.txt").read_text().splitlines()[:59]]
    got = [[float("%f" % v) for v in p] for p in poses]
    assert got == want


def test_untrusted_peer_is_validated(golden):
    """The scan length comes from the network: a short or long row must not reach the C side (which reads exactly nbeams
    floats from the buffer), and an out-of-range row index is answered with 404, not an exception in the handler."""
    wire = _wire()
    srv = wire.ScanServer(GOLDEN / "frames_head.csv", pandas_header=False)
    try:
        assert wire.fetch_scan("127.0.0.1", srv.port, 0, nbeams=1079).shape == (1079,)
        with pytest.raises(ValueError):
            wire.fetch_scan("127.0.0.1", srv.port, 0, nbeams=360)        # longer than expected: not silently truncated
        with pytest.raises(ValueError):
            wire.fetch_scan("127.0.0.1", srv.port, 0, nbeams=2000)       # shorter than expected
        for bad in (3, 9000, -1):
            with pytest.raises(ValueError):
                wire.fetch_scan("127.0.0.1", srv.port, bad, nbeams=1079)  # 404 from the server
        assert wire.fetch_scan("127.0.0.1", srv.port, 2, nbeams=1079).shape == (1079,)   # the server survived
    finally:
        srv.close()
