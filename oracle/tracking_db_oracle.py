"""CPU restatement of the track bookkeeping of the reference's TrackingDB
(backend/tracking/database.py:313-377 `add_frame` with its helpers :455-487).

TEST INFRASTRUCTURE ONLY: it checks that the records the front-end hands over
(slam_amd/backend/tracking/db_adapter.py) drive the reference's database to a
known state.  Pinned by tests/golden/ref_tracking_db.npz, which holds the state
of the reference's own TrackingDB after the same records were replayed into it
(oracle/make_golden.py).

State kept (names of the reference's fields in brackets):
  track_ids[f]      track id of every feature row of frame f, -1 if none
                    [frameId_to_trackIds_list]
  track_frames[t]   frames on which track t appears, in order [trackId_to_frames]
  links[(f, t)]     (x_left, x_right, y) of track t on frame f [linkId_to_link]
  leftover[f]       links of frame f's features that joined no track [leftover_links]
"""
from __future__ import annotations

from typing import Dict, List, Optional, Tuple

import numpy as np

NO_ID = -1


def _xyz(link) -> Tuple[float, float, float]:
    return (float(link.x_left), float(link.x_right), float(link.y))


class TrackDBOracle:
    def __init__(self):
        self.last_frame = -1
        self.last_track = -1
        self.track_ids: Dict[int, List[int]] = {}
        self.track_frames: Dict[int, List[int]] = {}
        self.links: Dict[Tuple[int, int], Tuple[float, float, float]] = {}
        self.leftover: Dict[int, list] = {}
        self.features: Dict[int, np.ndarray] = {}
        self._prev_links: Optional[list] = None

    # database.py:462-467
    def _new_link(self, frame: int, track: int, row: int, link) -> None:
        assert frame == self.last_frame or self.track_ids[frame][row] == NO_ID
        self.track_ids[frame][row] = track
        assert (frame, track) not in self.links
        self.links[(frame, track)] = _xyz(link)
        self.track_frames.setdefault(track, []).append(frame)

    # database.py:477-487
    def _drop_link_on_last_frame(self, track: int, row: int) -> None:
        del self.links[(self.last_frame, track)]
        self.track_ids[self.last_frame][row] = NO_ID
        assert self.track_frames[track][-1] == self.last_frame
        self.track_frames[track].pop()
        if not self.track_frames[track]:
            del self.track_frames[track]

    def add_frame(self, links, left_features, matches_to_previous_left=None, inliers=None) -> int:
        assert left_features.shape[0] == len(links)                       # :318-319
        prev = self.last_frame
        self.last_frame += 1
        cur = self.last_frame
        self.features[cur] = left_features
        self.track_ids[cur] = [NO_ID] * len(links)
        if cur == 0:                                                      # :325-328
            assert matches_to_previous_left is None
            self._prev_links = links
            return cur
        assert matches_to_previous_left is not None                       # :330
        if inliers is None:
            inliers = [True] * len(matches_to_previous_left)
        assert self.features[prev].shape[0] == len(matches_to_previous_left) == len(inliers)  # :332
        prev_ids = self.track_ids[prev]
        # best match so far of every current row: (distance, previous row, opened a new track)
        best: Dict[int, Tuple[float, int, bool]] = {}
        for m, ok in zip(matches_to_previous_left, inliers):              # :341-368
            if not ok:
                continue
            q, t, d = int(m.queryIdx), int(m.trainIdx), float(m.distance)
            held = best.get(t)
            if held is not None and held[0] <= d:                         # :348 keep the earlier, better one
                continue
            if held is not None:                                          # :350-359 undo the worse match
                track = prev_ids[held[1]]
                assert track == self.track_ids[cur][t]
                if held[2]:                                               # it had opened the track: close it
                    prev_ids[held[1]] = NO_ID
                    del self.links[(prev, track)]
                    assert self.track_frames[track].pop(0) == prev
                self._drop_link_on_last_frame(track, t)
            opened = prev_ids[q] == NO_ID                                 # :361
            best[t] = (d, q, opened)
            if opened:                                                    # :363-366
                self.last_track += 1
                self._new_link(prev, self.last_track, q, self._prev_links[q])
            self._new_link(cur, prev_ids[q], t, links[t])                 # :368
        self.leftover[prev] = [_xyz(l) for l, tid in zip(self._prev_links, prev_ids) if tid == NO_ID]  # :371-374
        self._prev_links = links
        return cur

    def all_frame_links(self, frame: int) -> np.ndarray:
        """(n,3) links of every feature row of `frame`, track member or not
        (database.py:199-218): what LoopClosure.frame_data reads per keyframe."""
        if frame == self.last_frame:
            return np.asarray([_xyz(l) for l in self._prev_links], dtype=np.float64).reshape(-1, 3)
        rest = iter(self.leftover[frame])
        return np.asarray([self.links[(frame, t)] if t != NO_ID else next(rest)
                           for t in self.track_ids[frame]], dtype=np.float64).reshape(-1, 3)

    # ---------------------------------------------------------------- summaries
    def summary(self) -> dict:
        """Arrays that describe the whole state (what the fixture stores)."""
        frames = sorted(self.track_ids)
        ids = np.concatenate([np.asarray(self.track_ids[f], dtype=np.int64) for f in frames]) if frames \
            else np.zeros(0, np.int64)
        off = np.concatenate([[0], np.cumsum([len(self.track_ids[f]) for f in frames])]).astype(np.int64)
        keys = sorted(self.links)
        return {
            "track_ids": ids, "track_ids_off": off,
            "last_track": np.int64(self.last_track), "last_frame": np.int64(self.last_frame),
            "link_keys": np.asarray(keys, dtype=np.int64).reshape(-1, 2),
            "link_xyz": np.asarray([self.links[k] for k in keys], dtype=np.float64).reshape(-1, 3),
            "track_len": np.asarray([len(self.track_frames[t]) for t in sorted(self.track_frames)], np.int64),
            "track_keys": np.asarray(sorted(self.track_frames), dtype=np.int64),
            "leftover_count": np.asarray([len(self.leftover.get(f, [])) for f in frames], dtype=np.int64),
        }
