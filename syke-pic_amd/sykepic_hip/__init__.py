"""MI355X-native CNN classification hot path of syke-pic (prob / train)."""

__version__ = "0.1.0"
