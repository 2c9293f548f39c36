# Builds the gfx950 shared library in-tree (fcd_amd/libfcd_hip.so) and the CPU oracle helpers.
HIPCC ?= /opt/rocm/bin/hipcc
ARCH ?= gfx950
SRCS := $(wildcard fcd_amd/csrc/*.hip)
OBJS := $(SRCS:.hip=.o)
HDRS := include/fcd_hip.h fcd_amd/csrc/fcd_common.h
FLAGS := --offload-arch=$(ARCH) -O3 -std=c++17 -fPIC -Wall -Wno-unused-function

all: fcd_amd/libfcd_hip.so

%.o: %.hip $(HDRS)
	$(HIPCC) $(FLAGS) -c $< -o $@

fcd_amd/libfcd_hip.so: $(OBJS)
	$(HIPCC) --offload-arch=$(ARCH) -shared -fPIC -o $@ $(OBJS)

clean:
	rm -f fcd_amd/csrc/*.o fcd_amd/libfcd_hip.so

.PHONY: all clean
