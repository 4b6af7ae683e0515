"""Scalar series of a training run, one JSON line per flush (build-only; the reference plots JPEGs, SURVEY.md 2 #10 - out
of scope).  `Series.add(name, value)` records a value for the current iteration, `flush()` writes the means since the last
flush, `tick()` advances the iteration."""
import json


class Series:
    def __init__(self, path=None, echo=print):
        self.path, self.echo, self.iteration, self._pending = path, echo, 0, {}

    def add(self, name, value):
        self._pending.setdefault(name, []).append(float(value))

    def tick(self):
        self.iteration += 1

    def flush(self):
        rec = {'iter': self.iteration}
        rec.update({k: sum(v) / len(v) for k, v in self._pending.items()})
        self._pending = {}
        if self.echo:
            self.echo(' '.join('%s=%s' % kv for kv in rec.items()))
        if self.path:
            with open(self.path, 'a') as f:
                f.write(json.dumps(rec) + '\n')
