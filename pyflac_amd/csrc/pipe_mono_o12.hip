#define PIPE_NAME mono_o12
#define PIPE_MS false
#define PIPE_NCH 1
#define PIPE_MAXO 12
#include "pipe_shape.inc"
