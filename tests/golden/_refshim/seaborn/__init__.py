"""Empty stand-in: the reference imports seaborn only for plotting (monitors.py), never on the hot path."""
