"""HTTP/JSON scan transport compatible with the reference's "LIDAR simulator" test rig.

Wire format restated from the reference (paths relative to its repository):
  * server  Submodule_2/Lidar_server.py:8-33 — ``GET /?param=<row>`` answers ``200 application/json`` with ONE
    HTTP chunk holding a JSON array of that row's values (``Transfer-Encoding: chunked``).  The rows come from
    ``pandas.read_csv``, i.e. the first CSV line is a header: ``param = N`` serves CSV line N + 1.
    (``param = 9000`` serves a PNG of the map there; not reproduced — no compute on that path.)
  * client  Submodule_2/esp32_edge.c:52-99 — requests ``?param=&param=<row>&pose_x=<%f>&pose_y=<%f>`` (the pose
    of the previous frame rides along as query parameters, the ESP32 relay logs it, ESP32_Server.c:35-54) and
    converts every JSON number with ``(float)value->valuedouble``.

This module is transport plumbing only (standard library, no third-party packages): `ScanServer` stands in for the
simulator, `fetch_scan` for the edge client, and `run_mapper_over_http` feeds the fetched frames to the engine's
device-resident frame loop (``slam_mapper_*``).  Nothing here computes on scans.
"""
from __future__ import annotations

import http.client
import json
import threading
from http.server import BaseHTTPRequestHandler, HTTPServer
from urllib.parse import parse_qs

import numpy as np


def _load_rows(csv_path, pandas_header=True):
    rows = []
    with open(csv_path) as f:
        for ln in f:
            ln = ln.strip()
            if ln:
                rows.append([float(v) for v in ln.rstrip(",").split(",")])
    return rows[1:] if pandas_header else rows   # pandas.read_csv consumes line 0 as the header (Lidar_server.py:5)


class ScanServer:
    """Serves scan frames in the reference simulator's format on 127.0.0.1:<port> (0 = ephemeral)."""

    def __init__(self, csv_path, port=0, pandas_header=True):
        rows = _load_rows(csv_path, pandas_header)
        log = self.requests = []

        class Handler(BaseHTTPRequestHandler):
            protocol_version = "HTTP/1.1"

            def log_message(self, *a):
                pass

            def do_GET(self):
                q = parse_qs(self.path[2:], keep_blank_values=False)
                try:
                    row = int(q.get("param", ["0"])[0])
                except ValueError:
                    row = -1
                log.append((row, q.get("pose_x", [None])[0], q.get("pose_y", [None])[0]))
                if not 0 <= row < len(rows):   # the reference would raise inside the handler; answer like a server
                    self.send_error(404, "no such scan row")
                    return
                body = json.dumps(rows[row]).encode()
                self.send_response(200)
                self.send_header("Content-type", "application/json")
                self.send_header("Connection", "keep-alive")
                self.send_header("Transfer-Encoding", "chunked")
                self.end_headers()
                self.wfile.write(hex(len(body))[2:].encode() + b"\r\n" + body + b"\r\n0\r\n\r\n")

        self.httpd = HTTPServer(("127.0.0.1", port), Handler)
        self.port = self.httpd.server_address[1]
        self.thread = threading.Thread(target=self.httpd.serve_forever, daemon=True)
        self.thread.start()

    def close(self):
        self.httpd.shutdown()
        self.httpd.server_close()


def fetch_scan(host, port, row, pose_x=0.0, pose_y=0.0, conn=None, nbeams=None):
    """One scan frame as float32, requested exactly like esp32_edge.c:59 does.  The peer is untrusted: with `nbeams`
    given, anything but a flat array of exactly that many numbers raises ValueError (the C side reads exactly
    nbeams floats from the buffer it is handed)."""
    own = conn is None
    if own:
        conn = http.client.HTTPConnection(host, port, timeout=10)
    conn.request("GET", "/?param=&param=%d&pose_x=%f&pose_y=%f" % (row, pose_x, pose_y))
    resp = conn.getresponse()
    raw = resp.read()   # http.client undoes the chunked framing
    if own:
        conn.close()
    if resp.status != 200:
        raise ValueError(f"scan row {row}: HTTP {resp.status}")
    data = json.loads(raw)
    r = np.asarray(data, np.float64).astype(np.float32)   # (float)valuedouble, esp32_edge.c:86
    if nbeams is not None and (r.ndim != 1 or r.size != nbeams):
        raise ValueError(f"scan row {row}: expected {nbeams} ranges, got an array of shape {r.shape}")
    return r


def run_mapper_over_http(pkg, engine, host, port, frames, nbeams=1079, angle_min=-2.351831, angle_inc=0.004363,
                         first_row=0):
    """Fetch `frames` scans over HTTP and run the engine's device-resident frame loop on them.
    Returns the poses [frames-1][3]."""
    import ctypes as C

    lib, conn = engine.lib, http.client.HTTPConnection(host, port, timeout=10)
    mp = C.c_void_p()
    engine._ck(lib.slam_mapper_create(engine.h, nbeams, angle_min, angle_inc, C.byref(mp)), "mapper_create")
    poses, pose = [], (C.c_float * 3)(0, 0, 0)
    try:
        r = np.ascontiguousarray(fetch_scan(host, port, first_row, conn=conn, nbeams=nbeams))
        engine._ck(lib.slam_mapper_first_frame(mp, r.ctypes.data_as(C.c_void_p)), "mapper_first_frame")
        for k in range(1, frames):
            r = np.ascontiguousarray(fetch_scan(host, port, first_row + k, pose[0], pose[1], conn=conn, nbeams=nbeams))
            engine._ck(lib.slam_mapper_next_frame(mp, r.ctypes.data_as(C.c_void_p), pose), "mapper_next_frame")
            poses.append([pose[0], pose[1], pose[2]])
    finally:
        lib.slam_mapper_destroy(mp)
        conn.close()
    return np.array(poses, np.float32)
