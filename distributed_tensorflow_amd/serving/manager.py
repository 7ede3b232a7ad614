"""Version management for servables.

The reference exports ``<output_dir>/<model_version>/`` directories
(recognition_flower/distributed/save_model.py:116-124,
recognition_flower/single/flower_train.py:195-203) -- the layout a model server
watches: every integer-named sub-directory of the model's base path is a
version, the highest one is served, and a new version that appears while the
server runs replaces the old one without dropping requests.

:class:`ModelManager` implements that policy for servable bundles: it scans the
base path, loads the newest complete version, keeps polling, and swaps the
loaded servable atomically (the new one is fully loaded and warmed before the
old one is released).  A version is complete once its ``saved_model.json``
manifest exists -- the exporter writes it last.
"""
from __future__ import annotations

import os
import threading
import time
from typing import Callable, Dict, List, Optional

from ..io.export import MANIFEST, Servable


def list_versions(base_path: str) -> List[int]:
    """Integer-named sub-directories holding a manifest, ascending."""
    out = []
    if os.path.isdir(base_path):
        for d in os.listdir(base_path):
            if d.isdigit() and os.path.exists(os.path.join(base_path, d, MANIFEST)):
                out.append(int(d))
    return sorted(out)


class ModelManager:
    def __init__(self, base_path: str, name: str = "default", device=None, poll_seconds: float = 2.0,
                 loader: Optional[Callable[[str], Servable]] = None, warmup: Optional[Callable] = None):
        self.base_path = base_path
        self.name = name
        self.device = device
        self.poll_seconds = float(poll_seconds)
        self._loader = loader or (lambda path: Servable(path, device=self.device))
        self._warmup = warmup
        self._lock = threading.Lock()
        self._servable: Optional[Servable] = None
        self._version: Optional[int] = None
        self._errors: Dict[int, str] = {}
        self._stop = threading.Event()
        self._thread: Optional[threading.Thread] = None
        self.loads = 0

    # ------------------------------------------------------------------ loading
    def refresh(self) -> bool:
        """Load the newest version if it is not the one being served.  Returns True when a swap happened."""
        versions = [v for v in list_versions(self.base_path) if v not in self._errors]
        if not versions:
            return False
        newest = versions[-1]
        if newest == self._version:
            return False
        try:
            servable = self._loader(os.path.join(self.base_path, str(newest)))
            if self._warmup is not None:
                self._warmup(servable)
        except Exception as e:  # noqa: BLE001 -- a broken export must not take the server down
            self._errors[newest] = "%s: %s" % (type(e).__name__, e)
            return False
        with self._lock:
            self._servable, self._version = servable, newest
        self.loads += 1
        return True

    def start(self) -> "ModelManager":
        self.refresh()
        if self.poll_seconds > 0:
            self._thread = threading.Thread(target=self._poll, name="model-poll", daemon=True)
            self._thread.start()
        return self

    def _poll(self):
        while not self._stop.wait(self.poll_seconds):
            self.refresh()

    def stop(self):
        self._stop.set()
        if self._thread is not None:
            self._thread.join(timeout=10)

    # ------------------------------------------------------------------ access
    def get(self, version: Optional[int] = None) -> Servable:
        with self._lock:
            servable, loaded = self._servable, self._version
        if servable is None:
            raise LookupError("no version of model %r is loaded" % self.name)
        if version is not None and int(version) != loaded:
            raise LookupError("version %s of model %r is not loaded (serving %s)" % (version, self.name, loaded))
        return servable

    @property
    def version(self) -> Optional[int]:
        return self._version

    def status(self) -> List[Dict]:
        """The ``model_version_status`` list of the model-status endpoint."""
        out = []
        if self._version is not None:
            out.append({"version": str(self._version), "state": "AVAILABLE",
                        "status": {"error_code": "OK", "error_message": ""}})
        for v, msg in sorted(self._errors.items()):
            out.append({"version": str(v), "state": "END", "status": {"error_code": "UNKNOWN", "error_message": msg}})
        return out


def wait_for_version(manager: ModelManager, timeout: float = 30.0) -> int:
    """Block until some version is loaded (used by the server at start-up and by tests)."""
    t0 = time.monotonic()
    while manager.version is None:
        if time.monotonic() - t0 > timeout:
            raise TimeoutError("no servable version appeared under %s" % manager.base_path)
        manager.refresh()
        if manager.version is None:
            time.sleep(0.1)
    return manager.version
