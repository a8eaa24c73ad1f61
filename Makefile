# Top-level build, target names as in the reference's Makefile (all / sharedlib / jni / clean).
#   make sharedlib   ec504_imageencoder_amd/libencoder.so  (hipcc gfx950 + gcc)  [+ ./libencoder.so link]
#   make all         ./encoder  — CLI equivalent of the reference's main.c (tools/encoder_cli.c)
#   make dropin      build/dropin_encoder — the reference's OWN main.c, unmodified, compiled where it lies
#                    against include/encoder.h and linked to libencoder.so (authoring container only)
#   make jni         libencoder_jni.so from the reference's encoder_jni.c (needs a JDK: JAVA_HOME)
#   make oracle      the CPU checker (test infrastructure, never linked into the product)
# stb_image.h (third-party single-header JPEG decoder the reference vendors) is taken from STB_DIR; it is
# compiled into the CALLER, exactly as in the reference, never into libencoder.so.
REF      ?= /root/reference
STB_DIR  ?= $(REF)/include
CC       ?= gcc
PKG      := ec504_imageencoder_amd
RPATH    := -Wl,-rpath,'$$ORIGIN/$(PKG)' -Wl,-rpath,'$$ORIGIN/../$(PKG)'

all: encoder

sharedlib:
	$(MAKE) -C $(PKG)/csrc
	ln -sf $(PKG)/libencoder.so libencoder.so

# stb_image.h is looked up AFTER this repository's include/ (-idirafter): "encoder.h" always resolves to ours, and
# when STB_DIR has no stb_image.h the callers are simply built without a JPEG loader.
STB_INC := -idirafter $(STB_DIR)

encoder: sharedlib tools/encoder_cli.c include/encoder.h
	$(CC) -O2 -w -Iinclude $(STB_INC) tools/encoder_cli.c -o $@ -L$(PKG) -lencoder $(RPATH) -lm

dropin: sharedlib
	@test -f $(REF)/main.c || { echo "$(REF)/main.c absent"; exit 1; }
	mkdir -p build
	$(CC) -g -w -Iinclude $(STB_INC) $(REF)/main.c -o build/dropin_encoder -L$(PKG) -lencoder $(RPATH) -lm

jni: sharedlib
	@test -n "$(JAVA_HOME)" -a -f "$(JAVA_HOME)/include/jni.h" || { echo "jni: needs a JDK (JAVA_HOME/include/jni.h)"; exit 1; }
	mkdir -p build/jni/include && ln -sf ../../../include/encoder.h build/jni/include/encoder.h
	$(CC) -g -w -fPIC -shared -Ibuild/jni -Iinclude $(STB_INC) -I$(JAVA_HOME)/include -I$(JAVA_HOME)/include/linux \
	    $(REF)/encoder_jni.c -o libencoder_jni.so -L$(PKG) -lencoder $(RPATH) -lm

oracle:
	$(MAKE) -C oracle

clean:
	$(MAKE) -C $(PKG)/csrc clean
	rm -rf build encoder libencoder.so libencoder_jni.so

.PHONY: all sharedlib dropin jni oracle clean
