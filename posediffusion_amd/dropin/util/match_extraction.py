"""Keypoint bookkeeping between hloc/COLMAP and the sampler (pose_diffusion/util/match_extraction.py:50-77).

`colmap_keypoint_to_pytorch3d` takes hloc's per-image keypoints (COLMAP pixel convention, 1-based image ids) and the
per-pair match index arrays, and returns the flat `(kp1, kp2, i12)` arrays `pd_ggs_set_matches` ingests: keypoints in
the pixel frame of the centre-cropped, resized images the sampler sees.  Here it is one table-wide transform plus one
gather (no per-image Python arithmetic): all keypoints are stacked once, every row carries its frame index, and the
matches become global row indices into that table.  Per element the arithmetic and its dtypes are the reference's
(`- 0.5` in the keypoints' own dtype, `- bbox` promoting to float64, `* scale`), so results are bit-identical
(tests/test_oracle_golden.py pins them against the reference file executed in place).  Unlike the reference it does
not overwrite the caller's `keypoints` dict.

Match extraction itself (`extract_match`, `run_hloc`: SuperPoint + SuperGlue through hloc / pycolmap) is outside this
engine's scope (third-party, needs network weights)."""
import numpy as np


def colmap_keypoint_to_pytorch3d(matches, keypoints, image_info):
    pairs = [(r, q, np.asarray(m)) for (r, q), m in matches.items() if m is not None]
    if not pairs:
        return None, None, None
    image_ids = list(keypoints)                                        # COLMAP ids, 1-based (:59)
    per_image = [np.asarray(keypoints[i]) for i in image_ids]
    first_row = dict(zip(image_ids, np.cumsum([0] + [len(k) for k in per_image[:-1]])))
    frame_of_row = np.repeat(np.asarray(image_ids, dtype=np.int64) - 1, [len(k) for k in per_image])
    table = np.concatenate(per_image, axis=0)
    crop_origin = np.asarray(image_info["bboxes_xyxy"])[frame_of_row, :2]
    zoom = np.asarray(image_info["resized_scales"])[frame_of_row]
    # COLMAP pixel centres -> OpenCV (:56), into the crop (:60), into the resized frame (:61)
    table = ((table - 0.5) - crop_origin) * zoom[:, None]
    rows1 = np.concatenate([first_row[r] + m[:, 0] for r, _, m in pairs])
    rows2 = np.concatenate([first_row[q] + m[:, 1] for _, q, m in pairs])
    i12 = np.repeat(np.array([[r - 1, q - 1] for r, q, _ in pairs]), [len(m) for _, _, m in pairs], axis=0)
    return table[rows1], table[rows2], i12


PRECOMPUTED_MATCHES = "pd_matches.npz"


def extract_match(image_paths=None, image_folder_path=None, image_info=None):
    """match_extraction.py:28-47 runs SuperPoint + SuperGlue through hloc / pycolmap -- third-party networks this engine
    does not replace.  Matches extracted elsewhere (the reference's own `extract_match` on any machine that has hloc)
    can be dropped next to the images as ``pd_matches.npz`` with the arrays demo.py:82-84 puts into matches_dict
    (kp1 [M,2], kp2 [M,2] pixel coordinates of the cropped + resized frames, i12 [M,2] frame indices); they are returned
    as extract_match would return them.  Without that file the call raises, as the missing dependency would."""
    import os
    if image_folder_path is not None:
        path = os.path.join(image_folder_path, PRECOMPUTED_MATCHES)
        if os.path.isfile(path):
            with np.load(path) as d:
                return (np.asarray(d["kp1"], dtype=np.float64), np.asarray(d["kp2"], dtype=np.float64),
                        np.asarray(d["i12"], dtype=np.int64))
    raise ImportError("extract_match needs hloc + pycolmap (SuperPoint / SuperGlue), which this engine does not replace: run "
                      "the reference's util/match_extraction.py where hloc is installed and either pass its result as "
                      f"matches_dict or save it as <image_folder>/{PRECOMPUTED_MATCHES} (kp1, kp2, i12)")
