"""Event-list ingest: timestamped DVS events -> count frames on the GPU.

The reference turns events into frames in one place, `create_images` of lens/collect_data.py:193-199: every
`timebin` milliseconds the events received since the last call become one `roi x roi` frame
(`frame[e.y-1, e.x-1] += 1`, stored as uint8, i.e. counts wrap modulo 256), written as
`frame_%05d.png`; a window without events writes nothing and does not advance the frame number.
Recorded streams come as text, one `t x y p` line per event after a header line with the sensor size,
usually inside a zip archive (the format lens/tools/dvstools.py:270-290 reads); this module reads that form,
the bare text file and a parquet table with the same columns, windows the stream the same way and
accumulates all windows with one `lens_accumulate_events` launch.  There is no CPU accumulation path.
"""
import os

import numpy as np
import torch

from .engine import accumulate_events


def load_events(path, skip_header=True):
    """A recorded event stream from disk: float64 t, int64 x, y, p.  By file type:
    `.zip`  -- the archive form lens/tools/dvstools.py:270-283 reads: the member `<archive stem>.txt` (or the
               only `.txt` member), first line = sensor size, then `t x y p` per line;
    `.parquet` -- columns t, x, y and optionally p (pyarrow);
    anything else -- the plain text form, see load_event_text."""
    low = path.lower()
    if low.endswith(".zip"):
        import io
        import zipfile
        with zipfile.ZipFile(path) as z:
            names = [n for n in z.namelist() if n.lower().endswith(".txt")]
            want = os.path.splitext(os.path.basename(path))[0] + ".txt"
            member = want if want in names else (names[0] if len(names) == 1 else None)
            if member is None:
                raise ValueError(f"{path}: expected the member {want} or exactly one .txt member, found {names}")
            with z.open(member) as fh:
                return load_event_text(io.TextIOWrapper(fh, encoding="ascii"), skip_header=skip_header)
    if low.endswith(".parquet"):
        import pyarrow.parquet as pq
        tab = pq.read_table(path)
        missing = [c for c in ("t", "x", "y") if c not in tab.column_names]
        if missing:
            raise ValueError(f"{path}: missing column(s) {missing} (need t, x, y and optionally p)")
        col = lambda n, dt: tab.column(n).to_numpy().astype(dt)
        x = col("x", np.int64)
        return col("t", np.float64), x, col("y", np.int64), (col("p", np.int64) if "p" in tab.column_names
                                                           else np.zeros_like(x))
    return load_event_text(path, skip_header=skip_header)


def load_event_text(path, skip_header=True):
    """`t x y p` per line (whitespace separated, t in the file's own unit); `path` is a file name or an open
    text file.  Returns float64 t, int64 x, y, p."""
    rows = np.loadtxt(path, dtype=np.float64, skiprows=1 if skip_header else 0, ndmin=2)
    if rows.shape[1] < 3:
        raise ValueError(f"{path}: expected at least 3 columns (t x y [p]), found {rows.shape[1]}")
    t = rows[:, 0]
    x = rows[:, 1].astype(np.int64)
    y = rows[:, 2].astype(np.int64)
    p = rows[:, 3].astype(np.int64) if rows.shape[1] > 3 else np.zeros_like(x)
    return t, x, y, p


def window_events(t, timebin_ms, t_unit_s=1e-6, t0=None):
    """Window index of every event and the start offsets of the non-empty windows.

    t must be non-decreasing (a recorded stream).  Window k covers [t0 + k*dt, t0 + (k+1)*dt) with
    dt = timebin_ms milliseconds expressed in the unit of t (`t_unit_s` seconds per tick; 1e-6 = microseconds).
    Returns (frame_off int64 [n_frames + 1], window_id int64 [n_frames]): events [off[i], off[i+1]) form
    frame i, which is window window_id[i]; empty windows produce no frame, like the reference."""
    t = np.asarray(t, dtype=np.float64)
    if t.size == 0:
        return np.zeros(1, dtype=np.int64), np.zeros(0, dtype=np.int64)
    if np.any(np.diff(t) < 0):
        raise ValueError("event timestamps must be non-decreasing")
    dt = float(timebin_ms) * 1e-3 / float(t_unit_s)
    if not dt > 0:
        raise ValueError("timebin must be positive")
    if abs(dt - round(dt)) <= 1e-9 * dt:      # 1e-3 / 1e-6 is not exactly 1000 in binary floating point
        dt = float(round(dt))
    start = t[0] if t0 is None else float(t0)
    if t[0] < start:
        raise ValueError("t0 lies after the first event")
    k = np.floor((t - start) / dt).astype(np.int64)
    first = np.flatnonzero(np.diff(k, prepend=k[0] - 1) != 0)          # first event of every non-empty window
    frame_off = np.concatenate([first, [t.size]]).astype(np.int64)
    return frame_off, k[first]


def frames_from_events(t, x, y, timebin_ms, roi_dim, t_unit_s=1e-6, t0=None, device="cuda", one_based=True):
    """uint8 frames [n_frames, roi_dim, roi_dim] on `device` and the window index of each frame.
    x, y are 1-based like the sensor's (collect_data.py:196) unless one_based=False."""
    frame_off, win = window_events(t, timebin_ms, t_unit_s, t0)
    x = np.asarray(x, dtype=np.int64)
    y = np.asarray(y, dtype=np.int64)
    if not one_based:
        x, y = x + 1, y + 1
    if x.size and (max(x.max(), y.max()) > 32767 or min(x.min(), y.min()) < -32768):
        raise IndexError("event coordinate does not fit the sensor's 16-bit range")
    dev = torch.device(device)
    xy = torch.from_numpy(np.stack([x, y], axis=1).astype(np.int16)).contiguous().to(dev)
    off = torch.from_numpy(frame_off).to(dev)
    frames = accumulate_events(xy, off, int(roi_dim))
    return frames, win


def save_frames(frames, folder, start_index=0):
    """`frame_%05d.png` like collect_data.py:198 (8-bit greyscale)."""
    from PIL import Image
    os.makedirs(folder, exist_ok=True)
    arr = frames.cpu().numpy()
    paths = []
    for i, fr in enumerate(arr):
        path = os.path.join(folder, f"frame_{start_index + i:05d}.png")
        Image.fromarray(fr, mode="L").save(path)
        paths.append(path)
    return paths


def main(argv=None):
    import argparse
    ap = argparse.ArgumentParser(description="event recording (t x y p as text, text inside a .zip, or .parquet) -> "
                                             "frame_%05d.png count frames, accumulated on the GPU")
    ap.add_argument("events")
    ap.add_argument("out_dir")
    ap.add_argument("--timebin", type=float, default=1000.0, help="window length in ms (LENS default 1000)")
    ap.add_argument("--roi_dim", type=int, default=80)
    ap.add_argument("--t_unit", type=float, default=1e-6, help="seconds per timestamp tick")
    ap.add_argument("--zero_based", action="store_true", help="coordinates start at 0 (recorded text files) instead of 1")
    ap.add_argument("--no_header", action="store_true")
    a = ap.parse_args(argv)
    t, x, y, _ = load_events(a.events, skip_header=not a.no_header)
    frames, win = frames_from_events(t, x, y, a.timebin, a.roi_dim, a.t_unit, one_based=not a.zero_based)
    paths = save_frames(frames, a.out_dir)
    print(f"{len(paths)} frames ({int(win[-1]) + 1 if len(win) else 0} windows) -> {a.out_dir}")


if __name__ == "__main__":
    main()
