"""Mirror of the reference's `primitive` module (primitive.rs:13)."""
