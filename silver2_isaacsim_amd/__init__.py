"""MI355X-native hydrodynamics force engine (drop-in for the per-body wrench
path of Joagai23/silver2_isaacsim).  See DESIGN.md."""
__version__ = "0.1.0"
