#!/usr/bin/env python3
"""Real-time blendshape inference -- drop-in for the reference's ``scripts/rt.py``.

Keeps the reference's I/O (scripts/rt.py:48-99, :175-238, :391-540):
  * ``RingBuffer(size)`` with ``write(np.ndarray)`` / ``read(n) -> Optional[np.ndarray]``;
  * PyAudio ``paFloat32`` mono capture via callback into a ``queue.Queue(maxsize=100)``, or synthetic
    ``randn * 0.01`` chunks with ``--no_audio``;
  * the CLI flags ``--model_path --config_path --sample_rate --target_fps --chunk_size --output_mode
    {udp,osc,file} --host --port --output_file --device --duration --no_audio``;
  * the output wire format: one UTF-8 ``json.dumps({"timestamp": <float>, "blendshapes": [52 floats]})``
    per frame as a UDP datagram or a JSONL line, or an OSC message to ``/blendshapes``;
  * pacing by sleeping to ``1 / target_fps``.

What runs behind it is the dual-stream production model on the GPU (the reference wires this script
to its legacy ``KoeMorphModel`` with a call that raises ``TypeError``, scripts/rt.py:370-372 vs
src/model/gaussian_face.py:278-283): each ``frame_samples`` chunk goes into the model's 8.5 s
sliding-window front end and one 52-coefficient frame comes out once the window is full.
"""
from __future__ import annotations

import argparse
import json
import logging
import queue
import socket
import sys
import time
from pathlib import Path
from typing import Optional, Sequence

import numpy as np

from ..wire import format_frames_raw
import torch

try:  # optional, exactly as in the reference (:24-36)
    import pyaudio
    HAS_PYAUDIO = True
except ImportError:  # pragma: no cover
    HAS_PYAUDIO = False
try:
    from pythonosc import udp_client
    HAS_OSC = True
except ImportError:  # pragma: no cover
    HAS_OSC = False

logging.basicConfig(level=logging.INFO)
logger = logging.getLogger(__name__)


class RingBuffer:
    """FIFO of mono float32 samples (reference scripts/rt.py:48-99): ``write`` silently drops what does
    not fit, ``read(n)`` returns a copy and consumes, or None when fewer than n samples are available."""

    def __init__(self, size: int):
        self.size = size
        self.buffer = np.zeros(size, dtype=np.float32)
        self.write_ptr = 0
        self.read_ptr = 0
        self.available = 0

    def write(self, data: np.ndarray):
        data = np.asarray(data).astype(np.float32)
        n = min(len(data), self.size - self.available)
        if n == 0:
            return
        end = self.write_ptr + n
        if end <= self.size:
            self.buffer[self.write_ptr:end] = data[:n]
        else:
            first = self.size - self.write_ptr
            self.buffer[self.write_ptr:] = data[:first]
            self.buffer[:n - first] = data[first:n]
        self.write_ptr = end % self.size
        self.available = min(self.available + n, self.size)

    def read(self, size: int) -> Optional[np.ndarray]:
        if self.available < size:
            return None
        end = self.read_ptr + size
        if end <= self.size:
            data = self.buffer[self.read_ptr:end].copy()
        else:
            first = self.size - self.read_ptr
            data = np.concatenate([self.buffer[self.read_ptr:], self.buffer[:size - first]])
        self.read_ptr = end % self.size
        self.available -= size
        return data


class AudioCapture:
    """PyAudio callback capture into a queue (reference :102-172)."""

    def __init__(self, sample_rate: int = 16000, chunk_size: int = 1024, channels: int = 1,
                 audio_queue: Optional[queue.Queue] = None):
        if not HAS_PYAUDIO:
            raise RuntimeError("PyAudio not available. Install with: pip install pyaudio")
        self.sample_rate = sample_rate
        self.chunk_size = chunk_size
        self.channels = channels
        self.audio_queue = audio_queue or queue.Queue()
        self.audio = pyaudio.PyAudio()
        self.stream = None
        self.is_recording = False

    def _audio_callback(self, in_data, frame_count, time_info, status):
        audio_data = np.frombuffer(in_data, dtype=np.float32)
        try:
            self.audio_queue.put_nowait(audio_data)
        except queue.Full:
            logger.warning("Audio queue full, dropping frame")
        return (None, pyaudio.paContinue)

    def start(self):
        if self.is_recording:
            return
        self.stream = self.audio.open(format=pyaudio.paFloat32, channels=self.channels, rate=self.sample_rate,
                                      input=True, frames_per_buffer=self.chunk_size,
                                      stream_callback=self._audio_callback)
        self.stream.start_stream()
        self.is_recording = True

    def stop(self):
        if not self.is_recording:
            return
        self.is_recording = False
        if self.stream:
            self.stream.stop_stream()
            self.stream.close()
            self.stream = None


class BlendshapeStreamer:
    """UDP / OSC / JSONL output (reference :175-238); wire format unchanged."""

    def __init__(self, output_mode: str = "udp", host: str = "127.0.0.1", port: int = 9001,
                 osc_address: str = "/blendshapes", output_file: Optional[str] = None):
        self.output_mode = output_mode
        self.host = host
        self.port = port
        self.osc_address = osc_address
        self.output_file = output_file
        if output_mode == "udp":
            self.socket = socket.socket(socket.AF_INET, socket.SOCK_DGRAM)
        elif output_mode == "osc":
            if not HAS_OSC:
                raise RuntimeError("python-osc not available. Install with: pip install python-osc")
            self.osc_client = udp_client.SimpleUDPClient(host, port)
        elif output_mode == "file":
            if output_file:
                self.file_handle = open(output_file, 'w')
            else:
                raise ValueError("output_file required for file mode")
        else:
            raise ValueError(f"Unknown output mode: {output_mode}")

    def send(self, blendshapes: np.ndarray, timestamp: float):
        """One frame; the JSON text is the reference's json.dumps byte for byte (koemorph_amd.wire, tested)."""
        if self.output_mode == "osc":
            self.osc_client.send_message(self.osc_address, blendshapes.tolist())
        else:
            self.send_batch(np.asarray(blendshapes, np.float32)[None, :], timestamp)

    def send_batch(self, frames: np.ndarray, timestamps, ports: Optional[Sequence[int]] = None):
        """A whole tick: frames (S, 52) -> S messages with the reference's wire format, encoded by ONE call into the C
        library instead of S json.dumps calls (30 720 per second at 1024 streams x 30 fps).  UDP: one datagram per
        row, to `ports[i]` if given (one receiver per stream) else to self.port; file: S JSONL lines; OSC: S messages."""
        frames = np.asarray(frames, np.float32)
        if self.output_mode == "osc":
            for row in frames:
                self.osc_client.send_message(self.osc_address, row.tolist())
            return
        raw, off = format_frames_raw(frames, timestamps, newline=self.output_mode == "file")
        if self.output_mode == "udp":
            view = memoryview(raw)
            for i in range(len(off) - 1):
                self.socket.sendto(view[off[i]:off[i + 1]], (self.host, ports[i] if ports is not None else self.port))
        else:
            self.file_handle.write(raw.decode("utf-8"))
            self.file_handle.flush()

    def close(self):
        if hasattr(self, 'socket'):
            self.socket.close()
        elif hasattr(self, 'file_handle'):
            self.file_handle.close()


class RealTimeInference:
    """Ring buffer in, one 52-vector per ``frame_samples`` out (reference :241-388)."""

    def __init__(self, model_path: Optional[str], config_path: Optional[str] = None, sample_rate: int = 16000,
                 target_fps: float = 30.0, buffer_duration: float = 2.0, device: str = "auto", model=None,
                 emotion_provider=None):
        self.sample_rate = sample_rate
        self.target_fps = target_fps
        self.frame_samples = int(sample_rate / target_fps)
        self.device = self._setup_device(device)
        self.audio_buffer = RingBuffer(int(buffer_duration * sample_rate))
        self.model = model if model is not None else self._load_model(model_path, config_path, emotion_provider)
        self.prev_blendshapes = None
        self.frame_count = 0

    def _setup_device(self, device: str) -> torch.device:
        if device == "auto":
            device = "cuda" if torch.cuda.is_available() else "cpu"
        dev = torch.device(device)
        if dev.type != "cuda":
            raise RuntimeError("koemorph_amd runs on the GPU only (no CPU fallback by design)")
        return dev

    def _load_model(self, model_path: str, config_path: Optional[str], emotion_provider):
        from ..model import SimplifiedDualStreamModel
        checkpoint = torch.load(model_path, map_location="cpu", weights_only=True)
        cfg = {}
        if config_path:
            import yaml
            with open(config_path) as f:
                cfg = (yaml.safe_load(f) or {}).get("model", {})
        elif isinstance(checkpoint.get("config"), dict):
            cfg = checkpoint["config"].get("model", checkpoint["config"])
        elif isinstance(checkpoint.get("model_config"), dict):
            cfg = checkpoint["model_config"]
        model = SimplifiedDualStreamModel(
            d_model=int(cfg.get("d_model", 256)), num_heads=int(cfg.get("num_heads", 8)),
            sample_rate=self.sample_rate, target_fps=int(self.target_fps),
            mel_sequence_length=int(cfg.get("mel_sequence_length", 256)), device=str(self.device),
            real_time_mode=True, emotion_provider=emotion_provider)
        state = checkpoint.get("model_state_dict", checkpoint)
        if any(k.startswith("audio_encoder.mel_encoder.") for k in state):
            # a checkpoint of the legacy multi-layer model (reference create_koemorph_model, scripts/rt.py:283-304).  The
            # reference's own loop cannot run it either: inference_step (scripts/rt.py:349-367) passes four positional
            # arguments, prosody features among them, to KoeMorphModel.inference_step(mel, emotion, prev).
            raise ValueError("this checkpoint holds the legacy multi-layer KoeMorphModel; load it with "
                             "koemorph_amd.model.create_koemorph_model(config) and drive inference_step(mel_features, "
                             "emotion_features, prev_blendshapes) directly -- scripts/rt.py serves the dual-stream model")
        model.load_state_dict(state)
        return model.to(self.device).eval()

    def process_audio_chunk(self, audio_chunk: np.ndarray):
        self.audio_buffer.write(audio_chunk)

    def inference_step(self) -> Optional[np.ndarray]:
        audio_data = self.audio_buffer.read(self.frame_samples)
        if audio_data is None:
            return None
        with torch.no_grad():
            bs = self.model.process_audio_frame_realtime(audio_data)
        if bs is None:
            return None
        self.prev_blendshapes = bs
        self.frame_count += 1
        return bs.cpu().numpy()

    def reset(self):
        self.prev_blendshapes = None
        self.frame_count = 0
        self.model.reset_realtime_state()


class AudioFileReader:
    """File playback source (reference scripts/rt_simplified.py:100-174, README.md:128-131 `--input_audio`): yields the
    file as chunks of `chunk_size` samples, the last one zero padded.  The reference sleeps chunk_duration between
    chunks to simulate real time; offline conversion (`--output_json`) does not need to."""

    def __init__(self, file_path: str, sample_rate: int = 16000, chunk_size: int = 1024):
        from scipy.io import wavfile                       # librosa / soundfile are not required for PCM / float WAV
        sr, data = wavfile.read(file_path)
        if data.ndim > 1:
            data = data.mean(axis=1)                       # mono=True
        if np.issubdtype(data.dtype, np.integer):
            data = data.astype(np.float32) / float(np.iinfo(data.dtype).max + 1)
        data = data.astype(np.float32)
        if sr != sample_rate:
            from scipy.signal import resample_poly
            from math import gcd
            g = gcd(int(sr), int(sample_rate))
            data = resample_poly(data, sample_rate // g, sr // g).astype(np.float32)
        self.audio_data, self.sample_rate, self.chunk_size = data, sample_rate, chunk_size
        self.current_pos = 0

    def __iter__(self):
        while self.current_pos < len(self.audio_data):
            chunk = self.audio_data[self.current_pos:self.current_pos + self.chunk_size]
            self.current_pos += len(chunk)
            if len(chunk) < self.chunk_size:
                chunk = np.pad(chunk, (0, self.chunk_size - len(chunk)), 'constant')
            yield chunk.astype(np.float32)

    def reset(self):
        self.current_pos = 0


def convert_file(inference: "RealTimeInference", input_audio: str, output_json: str, chunk_size: int = 1024) -> int:
    """`rt.py --input_audio a.wav --output_json out.jsonl` (README.md:128-131): run the real-time path over a file as
    fast as the GPU allows and write one JSONL line per output frame; timestamps are frame_index / target_fps."""
    streamer = BlendshapeStreamer(output_mode="file", output_file=output_json)
    frames, pending = 0, []
    # The sliding-window extractor skips a tick that arrives less than 0.3 update intervals after the last one by the WALL
    # clock (mel_sliding_window.py:267-269).  A file is converted faster than real time, so the gate is driven by AUDIO
    # time here -- the clock the reference's file player keeps by sleeping one chunk duration per chunk
    # (scripts/rt_simplified.py:100-174); without this the output would depend on how fast the GPU is.
    ticks = [0]
    extractor = getattr(inference.model, "mel_extractor", None)
    saved_clock = getattr(extractor, "_clock", None)
    if extractor is not None and saved_clock is not None:
        extractor._clock = lambda: 1.0 + ticks[0] * inference.frame_samples / float(inference.sample_rate)
    try:
        for chunk in AudioFileReader(input_audio, inference.sample_rate, chunk_size):
            inference.process_audio_chunk(chunk)
            # drain every whole frame the ring holds: inference_step also returns None while the 8.5 s context is still
            # filling, so "None" cannot end this loop (one read per 1024-sample chunk would let the 2 s ring overflow
            # and silently drop audio, scripts/rt.py:61-64)
            while inference.audio_buffer.available >= inference.frame_samples:
                ticks[0] += 1
                bs = inference.inference_step()
                if bs is None:
                    continue
                pending.append(np.asarray(bs, np.float32))
                if len(pending) == 256:
                    streamer.send_batch(np.stack(pending), (frames + np.arange(len(pending))) / inference.target_fps)
                    frames += len(pending)
                    pending = []
        if pending:
            streamer.send_batch(np.stack(pending), (frames + np.arange(len(pending))) / inference.target_fps)
            frames += len(pending)
    finally:
        streamer.close()
        if extractor is not None and saved_clock is not None:
            extractor._clock = saved_clock
    return frames


def build_parser() -> argparse.ArgumentParser:
    parser = argparse.ArgumentParser(description="Real-time KoeMorph inference")
    parser.add_argument("--model_path", type=str, required=True, help="Path to trained model checkpoint")
    parser.add_argument("--config_path", type=str, help="Path to model config file")
    parser.add_argument("--sample_rate", type=int, default=16000, help="Audio sample rate")
    parser.add_argument("--target_fps", type=float, default=30.0, help="Target blendshape frame rate")
    parser.add_argument("--chunk_size", type=int, default=1024, help="Audio chunk size for capture")
    parser.add_argument("--output_mode", type=str, default="udp", choices=["udp", "osc", "file"],
                        help="Output mode for blendshapes")
    parser.add_argument("--host", type=str, default="127.0.0.1", help="Output host")
    parser.add_argument("--port", type=int, default=9001, help="Output port")
    parser.add_argument("--output_file", type=str, help="Output file for file mode")
    parser.add_argument("--device", type=str, default="auto", help="Computation device")
    parser.add_argument("--duration", type=float, help="Duration to run (seconds), None for infinite")
    parser.add_argument("--no_audio", action="store_true", help="Disable audio capture (test mode)")
    # advertised by the reference's README.md:128-131 (its rt.py never implemented them)
    parser.add_argument("--input_audio", type=str, help="Convert this WAV file instead of capturing audio")
    parser.add_argument("--output_json", type=str, help="JSONL file written by --input_audio")
    return parser


def run_loop(inference: RealTimeInference, streamer: BlendshapeStreamer, args, audio_queue: queue.Queue,
             pace: bool = True) -> int:
    """The reference's main loop (:471-519).  Returns the number of frames sent."""
    start_time = time.time()
    frame_times = []
    sent = 0
    while True:
        loop_start = time.time()
        if args.duration and (time.time() - start_time) > args.duration:
            break
        processed_audio = False
        while not audio_queue.empty():
            try:
                inference.process_audio_chunk(audio_queue.get_nowait())
                processed_audio = True
            except queue.Empty:
                break
        if not processed_audio and args.no_audio:
            inference.process_audio_chunk(np.random.randn(args.chunk_size).astype(np.float32) * 0.01)
        blendshapes = inference.inference_step()
        if blendshapes is not None:
            streamer.send(blendshapes, time.time())
            sent += 1
        frame_time = time.time() - loop_start
        frame_times.append(frame_time)
        if len(frame_times) > 100:
            frame_times = frame_times[-100:]
        if pace:
            sleep_time = 1.0 / args.target_fps - frame_time
            if sleep_time > 0:
                time.sleep(sleep_time)
    if frame_times:
        logger.info(f"Average frame time: {np.mean(frame_times) * 1000:.1f}ms")
        logger.info(f"Processed {inference.frame_count} frames")
    return sent


def main(argv=None):
    args = build_parser().parse_args(argv)
    if not Path(args.model_path).exists():
        logger.error(f"Model file not found: {args.model_path}")
        return
    inference = RealTimeInference(model_path=args.model_path, config_path=args.config_path,
                                  sample_rate=args.sample_rate, target_fps=args.target_fps, device=args.device)
    if args.input_audio:
        n = convert_file(inference, args.input_audio, args.output_json or str(Path(args.input_audio).with_suffix(".jsonl")),
                         args.chunk_size)
        logger.info(f"Wrote {n} frames")
        return
    streamer = BlendshapeStreamer(output_mode=args.output_mode, host=args.host, port=args.port,
                                  output_file=args.output_file)
    audio_queue: queue.Queue = queue.Queue(maxsize=100)
    audio_capture = None
    if not args.no_audio and HAS_PYAUDIO:
        audio_capture = AudioCapture(sample_rate=args.sample_rate, chunk_size=args.chunk_size, audio_queue=audio_queue)
        audio_capture.start()
    try:
        run_loop(inference, streamer, args, audio_queue)
    except KeyboardInterrupt:
        logger.info("Interrupted by user")
    finally:
        if audio_capture:
            audio_capture.stop()
        streamer.close()


if __name__ == "__main__":
    main()
