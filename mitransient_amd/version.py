__version__ = "0.1.0"
reference_version = "1.3.0"      # mitransient release whose transient_path semantics are mirrored
