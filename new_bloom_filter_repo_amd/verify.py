"""Losslessness harness with the reference's result dictionaries:
verify_lossless  -- FixedVideoCompressor.verify_lossless (fixed_video_compressor.py:217-285)
verify_bit_exact -- verify_true_lossless.verify_bit_exact (verify_true_lossless.py:338-492),
                    without the OpenCV diagnostic image dumps.
Unlike the reference (which unwraps with `hasattr(x, 'data')` and then crashes on unequal plain
ndarrays, whose `.data` is a memoryview) both accept plain ndarrays and YUVFrame wrappers."""
import numpy as np


def _arr(frame):
    d = getattr(frame, "data", None)
    return d if isinstance(d, np.ndarray) else np.asarray(frame)


def verify_lossless(original_frames, decompressed_frames):
    if len(original_frames) != len(decompressed_frames):
        return {"lossless": False,
                "reason": f"Frame count mismatch: {len(original_frames)} vs {len(decompressed_frames)}",
                "avg_difference": float("inf")}
    exact, diff_frames, max_diff, max_diff_frame = 0, [], 0, -1
    for i, (o, d) in enumerate(zip(original_frames, decompressed_frames)):
        o, d = _arr(o), _arr(d)
        if np.array_equal(o, d):
            exact += 1
            continue
        frame_diff = np.mean(np.abs(o.astype(np.float32) - d.astype(np.float32)))
        diff_frames.append(i)
        if frame_diff > max_diff:
            max_diff, max_diff_frame = frame_diff, i
    ok = exact == len(original_frames)
    return {"lossless": ok, "exact_lossless": ok,
            "avg_difference": 0.0 if not diff_frames else max_diff,   # worst frame, as the reference reports it
            "max_difference": max_diff, "max_diff_frame": max_diff_frame,
            "exact_frame_matches": exact, "total_frames": len(original_frames), "diff_frames": diff_frames}


def verify_bit_exact(original_frames, decompressed_frames, color_space="BGR", verbose=False):
    if len(original_frames) != len(decompressed_frames):
        return {"success": False,
                "error": f"Frame count mismatch: {len(original_frames)} vs {len(decompressed_frames)}"}
    exact, diff_frames, details = 0, [], []
    for i, (o, d) in enumerate(zip(original_frames, decompressed_frames)):
        o, d = _arr(o), _arr(d)
        if o.shape != d.shape:
            diff_frames.append(i)
            details.append({"frame": i, "error": f"Shape mismatch: {o.shape} vs {d.shape}"})
            continue
        if np.array_equal(o, d):
            exact += 1
            continue
        diff_frames.append(i)
        diff = np.abs(o.astype(np.int16) - d.astype(np.int16))
        where = np.where(diff > 0)
        examples = []
        for j in range(min(10, len(where[0]))):
            c = tuple(axis[j] for axis in where)
            examples.append({"coordinates": str(c), "original_value": int(o[c]),
                             "decompressed_value": int(d[c]), "difference": int(diff[c])})
        details.append({"frame": i, "differences_found": len(where[0]), "examples": examples})
    result = {"success": exact == len(original_frames), "frames_compared": len(original_frames),
              "exact_matches": exact, "different_frames": len(diff_frames),
              "different_frame_indices": diff_frames, "diff_details": details}
    if verbose:
        print(f"Bit-exact verification: {'SUCCESS' if result['success'] else 'FAILED'}")
        print(f"  Exact frame matches: {exact}/{len(original_frames)}")
    return result
