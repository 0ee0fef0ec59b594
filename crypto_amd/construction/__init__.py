"""Mirror of the reference's `construction` module (construction.rs:19-22) for the hot path."""
