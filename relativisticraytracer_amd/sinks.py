"""Frame sinks for the headless driver (replaces the reference's ScreenRecorder,
src/main.cpp:29-124, which pipes glReadPixels() output into ffmpeg).

Frames arrive as (h, w, 4) uint8 arrays in the kernel's native order: RGBA, bottom-up rows
(raymarcher.cu:168) -- the same bytes glReadPixels hands the reference's recorder."""
import os
import shutil
import subprocess

import numpy as np


class RawSink:
    """Concatenated raw RGBA frames, bottom-up -- byte-for-byte what the reference writes into the
    ffmpeg pipe (main.cpp:85-97).  Convert later with the reference's own arguments:
    ffmpeg -f rawvideo -pix_fmt rgba -s WxH -r 24 -i frames.rgba -vf vflip -c:v libx264 ..."""

    def __init__(self, path, width, height):
        self.f = open(path, "wb")
        self.width, self.height, self.frames = width, height, 0

    def write(self, frame):
        frame = np.ascontiguousarray(frame, np.uint8)
        if frame.shape != (self.height, self.width, 4):
            raise ValueError("frame shape mismatch")
        self.f.write(frame.tobytes())
        self.frames += 1

    def close(self):
        self.f.close()


class PPMSink:
    """One binary PPM per frame (RGB, top-down: the vertical flip the reference delegates to
    ffmpeg's `-vf vflip`, main.cpp:67, is applied here)."""

    def __init__(self, directory, width, height, prefix="frame"):
        os.makedirs(directory, exist_ok=True)
        self.dir, self.prefix = directory, prefix
        self.width, self.height, self.frames = width, height, 0

    def write(self, frame):
        frame = np.ascontiguousarray(frame, np.uint8)
        if frame.shape != (self.height, self.width, 4):
            raise ValueError("frame shape mismatch")
        self.frames += 1
        with open(os.path.join(self.dir, f"{self.prefix}_{self.frames:05d}.ppm"), "wb") as f:
            f.write(b"P6\n%d %d\n255\n" % (self.width, self.height))
            f.write(frame[::-1, :, :3].tobytes())

    def close(self):
        pass


class FFmpegSink:
    """The reference's recorder, argument for argument (main.cpp:60-72).  Only usable where an
    `ffmpeg` binary exists; raises otherwise (there is none in the build image)."""

    def __init__(self, path, width, height, fps=24):
        exe = shutil.which("ffmpeg")
        if exe is None:
            raise RuntimeError("ffmpeg not found in PATH (the reference prints the same complaint, main.cpp:76)")
        cmd = [exe, "-y", "-f", "rawvideo", "-pix_fmt", "rgba", "-s", f"{width}x{height}", "-r", str(fps),
               "-i", "-", "-vf", "vflip", "-c:v", "libx264", "-preset", "fast", "-crf", "18",
               "-pix_fmt", "yuv420p", path]
        self.p = subprocess.Popen(cmd, stdin=subprocess.PIPE, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
        self.width, self.height, self.frames = width, height, 0

    def write(self, frame):
        self.p.stdin.write(np.ascontiguousarray(frame, np.uint8).tobytes())
        self.frames += 1

    def close(self):
        self.p.stdin.close()
        self.p.wait()


def open_sink(spec, width, height, fps=24):
    """spec: None | 'x.rgba' | 'dir/' (PPM per frame) | 'x.mp4' (ffmpeg)."""
    if not spec:
        return None
    if spec.endswith(".rgba") or spec.endswith(".raw"):
        return RawSink(spec, width, height)
    if spec.endswith(".mp4"):
        return FFmpegSink(spec, width, height, fps)
    return PPMSink(spec, width, height)
