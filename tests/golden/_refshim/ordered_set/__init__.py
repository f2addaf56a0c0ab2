"""Stand-in for the `ordered_set` wheel (absent here): insertion-ordered, de-duplicating iterable.
The reference uses it once, to de-duplicate parameters handed to Adam (solvers.py:182)."""


class OrderedSet:
    def __init__(self, iterable=()):
        self._d = dict.fromkeys(iterable)

    def __iter__(self):
        return iter(self._d)

    def __len__(self):
        return len(self._d)

    def __contains__(self, item):
        return item in self._d
