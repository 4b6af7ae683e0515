"""tflib.plot - same plot/tick/flush API as TF/tflib/plot.py:11-41, writing JSON lines instead of
matplotlib JPEGs (no display stack on the GPU boxes)."""
import collections
import json

_since_beginning = collections.defaultdict(lambda: {})
_since_last_flush = collections.defaultdict(lambda: {})
_iter = [0]
log_path = None


def tick():
    _iter[0] += 1


def plot(name, value):
    _since_last_flush[name][_iter[0]] = float(value)


def flush():
    prints = []
    rec = {'iter': _iter[0]}
    for name, vals in _since_last_flush.items():
        mean = sum(vals.values()) / max(len(vals), 1)
        prints.append('{}\t{}'.format(name, mean))
        rec[name] = mean
        _since_beginning[name].update(vals)
    print('iter {}\t{}'.format(_iter[0], '\t'.join(prints)))
    _since_last_flush.clear()
    if log_path:
        with open(log_path, 'a') as f:
            f.write(json.dumps(rec) + '\n')
