"""tombo_amd: MI355X-native resquiggle engine (drop-in for tombo.resquiggle.resquiggle_read)."""
__version__ = '0.1.0'
