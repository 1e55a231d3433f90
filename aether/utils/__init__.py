"""Import-path alias of the reference's `aether.utils` for the functions on (or next to) the accelerated path."""
